"""-m gpu: site-sharding on one GPU through pf_forward_shards_emulated (the test backend:
real kernels and per-shard workspaces, the two RCCL all-reduces replaced by device sums)."""
import os

import numpy as np
import pytest

from phyloformer_amd.msa_sim import simulate_batch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("nshards", [2, 3, 8])
def test_shards_match_unsharded_and_reference(engines, golden, nshards):
    g = golden("configs.npz")
    e = engines("pf")
    a = g["c2_idx"]                                   # 3 x (20 x 200)
    full = e.forward(a)
    sh = e.forward_shards_emulated(a, nshards)
    assert np.abs(sh - full).max() <= 2e-5
    assert np.abs(sh - g["c2_dist"]).max() <= 1e-4


@pytest.mark.parametrize("precise", [0, -1])
def test_ragged_and_empty_shards(engines, precise):
    # precise = 0: the default kernels' shard edge cases; -1: as routed (both shapes take the float64 path)
    e = engines("pf_indel", precise=precise)
    idx = simulate_batch(2, 6, 37, seed=5, gaps=True)  # 37 sites over 8 ranks: 5,5,...,2
    full = e.forward(idx)                              # 6 sequences: distances reach ~6 here
    assert np.abs(e.forward_shards_emulated(idx, 8) - full).max() <= 2e-5 * max(1.0, float(full.max()))
    tiny = simulate_batch(1, 5, 3, seed=6)             # 3 sites over 4 ranks: last rank empty
    ft = e.forward(tiny)
    # 5 sequences x 3 sites is ill-conditioned (|x| ~ 400, logits ~ 160; the fp32 numpy oracle is
    # already 2e-5 away from its own fp64 evaluation): this case only checks that 1-site shards and
    # an empty rank are handled, with a bound that any indexing error (O(0.1)) would break
    assert np.abs(e.forward_shards_emulated(tiny, 4) - ft).max() <= 5e-4


def test_config4_60x2000_eight_shards(engines, repo):
    path = os.path.join(repo, "tests/golden/configs_big.npz")
    if not os.path.exists(path):
        pytest.skip("configs_big.npz not generated")
    g = np.load(path)
    got = engines("pf").forward_shards_emulated(g["c4_idx"], 8)   # 8 x 250 sites, BASELINE configs[3]
    err = np.abs(got - g["c4_dist"]).max()
    print(f"60x2000 over 8 emulated ranks: max-abs error vs reference {err:.3e}")
    assert err <= 1e-4


def test_rccl_single_rank_communicator(weights, golden):
    """dlopen(librccl), ncclGetUniqueId/CommInitRank/AllReduce on the engine's two streams, each with its OWN
    single-rank communicator (pf_comm_init creates two from the 256-byte id blob): the collectives of a sharded
    forward (7 per half-batch, two halves for this batch of two) must leave the result unchanged."""
    from phyloformer_amd.engine import Engine
    g = golden("configs.npz")
    a = g["c2_idx"][:2]
    with Engine(weights("pf"), 0) as e:
        ref = e.forward(a)
        e.set_option("force_rccl", 1)
        e.comm_init(e.unique_id(), 0, 1)
        e.set_option("profile", 1)
        e.profile_reset()
        got = e.forward_sharded(a, 0, 200, 200)
        n, _ms = e.profile_get("allreduce")
        assert e.collective_count() == n          # the always-on counter agrees with the event-bracketed one
        info = e.comm_info()
        # the plain entry points never communicate, communicator or not (alignment-level data parallelism
        # on a handle that also serves site-sharded calls)
        e.profile_reset()
        plain = e.forward(a)
        n_plain, _ms = e.profile_get("allreduce")
        e.comm_destroy()
    assert n == 14 and n_plain == 0
    assert np.array_equal(got, ref) and np.array_equal(plain, ref)
    assert info["library"].endswith(".so.1") and info["version"] > 20000
    print("RCCL:", info)


def test_range_recheck_of_a_sharded_call_runs_the_float64_collectives(weights):
    """The range re-check (option "recheck_above", tests/test_gpu_precise.py) inside pf_forward_sharded over a real
    single-rank communicator pair: the batch of three takes the default kernels' 14 collectives, then the one alignment
    whose distances exceed 8 takes the float64 forward's 7 - the same alignments on every rank, since all ranks hold the
    same all-reduced result - and the call returns pf_forward's bits."""
    from phyloformer_amd.engine import Engine
    n, l = 33, 33
    sat = np.random.default_rng(805854907).integers(0, 22, (1, n, l)).astype(np.uint8)
    sim = simulate_batch(2, n, l, seed=5)
    batch = np.concatenate([sim[:1], sat, sim[1:]])
    with Engine(weights("pf_selreg"), 0) as e:
        ref = e.forward(batch)
        assert e.rechecked_count() == 1 and float(ref[1].max()) > 8.0 and float(ref[[0, 2]].max()) < 8.0
        e.set_option("force_rccl", 1)
        e.comm_init(e.unique_id(), 0, 1)
        e.profile_reset()
        got = e.forward_sharded(batch, 0, l, l)
        assert e.collective_count() == 14 + 7 and e.rechecked_count() == 1
        e.set_option("recheck_above", 0)
        e.profile_reset()
        raw = e.forward_sharded(batch, 0, l, l)
        assert e.collective_count() == 14 and e.rechecked_count() == 0
        e.comm_destroy()
    assert np.array_equal(got, ref)
    assert np.array_equal(raw[[0, 2]], ref[[0, 2]]) and not np.array_equal(raw[1], ref[1])


def test_partial_site_range_without_communicator_is_refused(weights, golden):
    """pf_forward_sharded with fewer sites than L_total and no communicator would return partial sums divided
    by L_total: PF_ESTATE instead."""
    from phyloformer_amd.engine import Engine, EngineError
    a = golden("configs.npz")["c2_idx"][:1]
    with Engine(weights("pf"), 0) as e:
        with pytest.raises(EngineError) as exc:
            e.forward_sharded(a[:, :, :120], 0, 120, 200)
        assert "communicator" in str(exc.value)
        assert np.isfinite(e.forward(a)).all()          # the handle stays usable


def test_overlapped_half_batches_equal_serial_schedule(weights, golden):
    """Site-sharded forwards with >= 2 alignments run as two half-batches on two streams so that the
    all-reduce of one half overlaps the compute of the other (option "overlap", default on).  With a real
    (single-rank) RCCL communicator: bit-identical to the serial schedule, 2 x (n_blocks + 1) = 14
    collectives instead of 7, and an odd batch splits 2 + 1."""
    from phyloformer_amd.engine import Engine
    a = golden("configs.npz")["c2_idx"]                   # 3 x (20 x 200)
    with Engine(weights("pf"), 0) as e:
        ref = e.forward(a)
        e.set_option("force_rccl", 1)
        e.comm_init(e.unique_id(), 0, 1)
        out, ncoll = {}, {}
        for overlap in (0, 1):
            e.set_option("overlap", overlap)
            e.set_option("profile", 1)
            e.profile_reset()
            out[overlap] = e.forward_sharded(a, 0, 200, 200)
            ncoll[overlap] = e.profile_get("allreduce")[0]
            e.set_option("profile", 0)
            again = e.forward_sharded(a, 0, 200, 200)     # un-profiled: the two streams really run concurrently
            assert np.array_equal(again, out[overlap])
        single = e.forward_sharded(a[:1], 0, 200, 200)    # one alignment: nothing to split
        e.comm_destroy()
    assert ncoll == {0: 7, 1: 14}
    assert np.array_equal(out[0], ref) and np.array_equal(out[1], ref)
    assert np.array_equal(single, ref[:1])


def test_debug_taps_cover_the_whole_batch_when_collectives_run(weights, golden):
    """with debug_keep a site-sharded forward must not be cut in two halves (each tap name holds
    one tensor): 7 collectives, taps of the full batch, same distances."""
    from phyloformer_amd.engine import Engine
    a = golden("configs.npz")["c2_idx"][:2]
    with Engine(weights("pf"), 0) as e:
        ref = e.forward(a)
        e.set_option("force_rccl", 1)
        e.comm_init(e.unique_id(), 0, 1)
        e.set_option("debug_keep", 1)
        e.profile_reset()
        got = e.forward_sharded(a, 0, 200, 200)
        assert e.collective_count() == 7
        assert e.debug_read("srow0").size == 2 * 190 * 72 and e.debug_read("x6").size == 2 * 190 * 200 * 64
        e.set_option("debug_keep", 0)
        e.comm_destroy()
    assert np.array_equal(got, ref)


def test_chunk_budget_counts_both_workspaces(weights, golden):
    """the per-chunk budget (ws_limit_mb) bounds ws + ws2 together and uses the real column-statistics
    plan; a batch chunked under a tight budget returns the bits of the unchunked run, on one and on two streams."""
    from phyloformer_amd.engine import Engine
    a = np.concatenate([golden("configs.npz")["c2_idx"]] * 4)[:11]          # 11 alignments of 20 x 200
    with Engine(weights("pf"), 0) as e:
        whole = e.forward(a)
        for two in (1, 0, 1):
            e.set_option("two_streams", two)
            e.set_option("ws_limit_mb", 48)
            assert np.array_equal(e.forward(a), whole)
            e.set_option("ws_limit_mb", 24576)


def test_empty_rank_joins_every_collective_with_zeros(weights, golden):
    """the branch a rank without sites takes (L_total < world; `l_begin == l_end`) is the one
    function real ranks run.  A single-rank `force_rccl` communicator takes it on one GPU: the same two halves on the
    same two streams / communicators as its peers would issue - 14 collectives for a batch of 3 (7 with overlap off,
    7 for a lone alignment) - and the rank's contribution, hence here the result, is all zeros.  The handle is a
    working engine afterwards (`reducing` / `cur` are restored whatever way the call leaves)."""
    from phyloformer_amd.engine import Engine
    a = golden("configs.npz")["c2_idx"]                    # 3 x (20 x 200)
    empty = np.zeros((3, 20, 0), np.uint8)
    with Engine(weights("pf"), 0) as e:
        ref = e.forward(a)
        with pytest.raises(ValueError):                    # no communicator: an empty site range is just bad dims
            e.forward_sharded(empty, 200, 200, 200)
        e.set_option("force_rccl", 1)
        e.comm_init(e.unique_id(), 0, 1)
        for overlap, B, want in ((1, 3, 14), (0, 3, 7), (1, 1, 7)):
            e.set_option("overlap", overlap)
            e.profile_reset()
            got = e.forward_sharded(empty[:B], 200, 200, 200)
            assert e.collective_count() == want, (overlap, B, e.collective_count())
            assert got.shape == (B, 190) and not got.any()
        # device entry: the same branch, asynchronous
        d_out = e.malloc(3 * 190 * 4)
        e.h2d(d_out, np.ones((3, 190), np.float32))
        e.set_option("overlap", 1)
        e.profile_reset()
        e.forward_sharded_device(0, 3, 20, 200, 200, 200, d_out)
        back = np.empty((3, 190), np.float32)
        e.d2h(back, d_out)
        e.free(d_out)
        assert e.collective_count() == 14 and not back.any()
        with pytest.raises(ValueError):                    # the plain entry points never take that branch
            e.forward(empty)
        assert np.array_equal(e.forward_sharded(a, 0, 200, 200), ref)       # full range again: a normal forward
        e.comm_destroy()
        assert np.array_equal(e.forward(a), ref)
