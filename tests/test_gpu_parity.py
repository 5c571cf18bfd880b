"""-m gpu: the HIP path (through the C ABI) against the oracle and the reference goldens.

Tolerance: BASELINE.json's north star asks for <= 1e-4 max-abs on the predicted
distances against the reference CPU forward.  The split-fp16 MFMA scheme (bf16 until round 5) lands
around 1e-5, so the tests assert tighter bounds where the data allow it.
"""
import glob
import os

import numpy as np
import pytest

from oracle import pf_oracle as O
from phyloformer_amd.fasta import load_alignment
from phyloformer_amd.msa_sim import simulate_batch

import devmath

pytestmark = pytest.mark.gpu
TOL = 1e-4   # north-star bound on distances


def test_hardware_layout_selftest(engines):
    """Cross-lane primitives and the MFMA operand/result layout the kernels assume."""
    out = engines("pf").selftest()
    lane = np.arange(64)
    assert np.array_equal(out[0:64], (lane % 32) * 2 + 32.0)                 # pair_sum
    assert np.array_equal(out[64:128], (lane ^ 32).astype(np.float32))        # pair_other
    rows = lane.reshape(4, 16).sum(1)
    assert np.array_equal(out[128:192], np.repeat(rows, 16).astype(np.float32))
    halves = lane.reshape(2, 32).sum(1)
    assert np.array_equal(out[192:256], np.repeat(halves, 32).astype(np.float32))
    d1 = out[256:1280].reshape(64, 16)
    d2 = out[1280:2304].reshape(64, 16)
    for l in range(64):
        for r in range(16):
            m, n = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), l & 31
            assert d1[l, r] == m + 32 * ((n % 16) % 8), (l, r)
            assert d2[l, r] == n % 16, (l, r)


_ERRORS = {}


def _record(case, got, want, f64=None):
    """Absolute max error of one parity case; the table is printed and, on the GPU box, written to
    gpurun_out/parity_errors.json (copied into DESIGN.md section 5).  ``f64``: the same forward evaluated in
    float64 by the oracle - the GPU's and the fp32 reference's distance from it say how much of the error is
    the fp32 arithmetic of the comparison target itself."""
    err, scale = float(np.abs(got - want).max()), float(np.abs(want).max())
    _ERRORS[case] = {"max_abs_err": err, "max_abs_ref": scale}
    extra = ""
    if f64 is not None:
        _ERRORS[case]["gpu_vs_fp64"] = float(np.abs(got - f64).max())
        _ERRORS[case]["fp32_reference_vs_fp64"] = float(np.abs(want - f64).max())
        extra = (f"; vs the fp64 evaluation: GPU {_ERRORS[case]['gpu_vs_fp64']:.3e}, "
                 f"fp32 reference {_ERRORS[case]['fp32_reference_vs_fp64']:.3e}")
    print(f"parity {case}: max-abs error {err:.3e} (max |reference| {scale:.3g}){extra}")
    try:
        import json
        os.makedirs("gpurun_out", exist_ok=True)
        with open(os.path.join("gpurun_out", "parity_errors.json"), "w") as fh:
            json.dump(_ERRORS, fh, indent=1, sort_keys=True)
    except OSError:
        pass
    return err, scale


def _check(err, scale, what=None):
    """The north star's bound: 1e-4 ABSOLUTE on the distances, for every case - in distribution (errors of
    1e-5 and below) and out of it (2-5 sequences or uniformly random residues drive distances to 5-12 and the
    error to 9e-5 at worst; the fp32 reference itself is 1e-5 from an fp64 evaluation there, see the table).
    A case that crosses 1e-4 is a finding about the kernels, not a tolerance to widen.  The second assertion is
    relative and only bites where distances are small: 5e-5 of max(1, largest distance)."""
    assert err <= TOL, (what, err)
    assert err <= 5e-5 * max(1.0, scale), (what, err, scale)


def test_tiny_taps_localise_every_kernel(engines, weights, golden):
    g = golden("taps_tiny.npz")
    w = weights("pf").tensors
    e = engines("pf", precise=0)        # the default kernels' taps (5 x 16 would otherwise take the float64 path)
    e.set_option("debug_keep", 1)
    try:
        # per-buffer bounds (relative to the largest reference entry): 3-4 x the measured errors on this
        # adversarial input (worst: srow 2.4e-5, mrow 2.5e-5, ctx 8.6e-5, x 3.2e-5 in block 5, error table in
        # DESIGN.md section 5) and far below what any layout / indexing bug (O(1) errors) produces
        d = e.forward(g["idx"])
        P, L = 10, 16
        x0 = e.debug_read("x0").reshape(P, L, 64)
        assert np.array_equal(x0, g["embed"])                               # bit-exact gather + add
        x_in = g["embed"]
        for k in range(6):
            srow = e.debug_read(f"srow{k}").reshape(P, 72)
            want, _q = devmath.expected_srow(w, k, x_in)
            tap_rel = {}

            def close(name, got, wnt, rel):
                # measured margins go into the error table (relative to the largest reference entry)
                tap_rel[name] = float(np.abs(got - wnt).max() / np.abs(wnt).max())
                assert tap_rel[name] <= rel, (name, tap_rel[name])

            close(f"srow{k}", srow, want, 1e-4)
            mrow = e.debug_read(f"mrow{k}").reshape(P, 4, 64)
            wantm = devmath.expected_mrow(w, k, want, L)[:, :4]      # (row 4, the bias, is no longer stored per pair)
            close(f"mrow{k}", mrow, wantm, 1e-4)
            ctx = e.debug_read(f"ctx{k}").reshape(L, 64)
            wantc, _qc = devmath.expected_ctx(w, k, g[f"block{k}.row"])
            close(f"ctx{k}", ctx, wantc, 3e-4)
            xk = e.debug_read(f"x{k + 1}").reshape(P, L, 64)
            ref = g[f"block{k}.ffn"]
            close(f"x{k + 1}", xk, ref, 1e-4)
            _ERRORS[f"tiny_taps block {k}: relative error of srow / mrow / ctx / x"] = {
                "max_abs_err": max(tap_rel.values()), "max_abs_ref": None,
                "detail": {n: round(v, 9) for n, v in tap_rel.items()}}
            x_in = ref
        # uniform-random residues incl. X and gaps drive |x| to ~130 and distances to ~12
        err, scale = _record("tiny_taps 5x16 random residues", d, g["dist"],
                             f64=O.forward(w, g["idx"], dtype=np.float64))
        _check(err, scale, "tiny_taps")
    finally:
        e.set_option("debug_keep", 0)


@pytest.mark.parametrize("precise", [0, -1])
def test_oracle_parity_small_shapes(engines, weights, precise):
    """Seeded synthetic alignments incl. ragged tile tails, gaps and the minimum sizes - through the default kernels
    (precise = 0: their edge cases) and as the product routes them (-1: all but the last - fewer than 32 sites or 8,192 tokens - take the float64 path)."""
    e = engines("pf_indel", precise=precise)
    w = weights("pf_indel").tensors
    for (n, l, gaps, seed) in [(2, 1, False, 1), (3, 31, False, 2), (4, 32, True, 3), (5, 33, True, 4),
                               (7, 65, True, 5), (12, 100, False, 6)]:
        idx = simulate_batch(2, n, l, seed=seed, gaps=gaps)
        got = e.forward(idx)
        want = O.forward_batch(w, idx)
        assert got.shape == want.shape == (2, n * (n - 1) // 2)
        # 2-5 sequences is far outside the training distribution: the residual stream reaches
        # |x| ~ 560 and logits ~ 190
        err, scale = _record(f"oracle {n}x{l}{' gapped' if gaps else ''} pf_indel{'' if precise else ' (default kernels)'}",
                             got, want, f64=O.forward_batch(w, idx, dtype=np.float64))
        _check(err, scale, (n, l))


def test_oracle_parity_shapes_around_the_kernels_block_sizes(engines, weights):
    """Site counts around the 32-site tile, the 64-site residue block of k_embed and the 16 / 32 / 64-pair
    runs of k_colstats (one group, several groups, a short last group), batched and alone - the lone
    alignment takes the run-sized column-statistics blocks, the batch the group-sized ones.  Default kernels on
    every shape (33 x 12 would otherwise take the float64 path)."""
    e = engines("pf", precise=0)
    w = weights("pf").tensors
    cases = [(9, 63, 1, 11), (9, 64, 2, 12), (9, 65, 1, 13), (6, 127, 3, 14), (6, 129, 1, 15), (17, 40, 1, 16),
             (17, 40, 4, 17), (24, 33, 1, 18), (33, 12, 1, 19), (40, 70, 1, 20)]
    for (n, l, b, seed) in cases:
        idx = simulate_batch(b, n, l, seed=seed, gaps=(seed % 3 == 0))
        got = e.forward(idx)
        want = O.forward_batch(w, idx)
        err, scale = _record(f"oracle {n}x{l} batch {b} pf", got, want,
                             f64=O.forward_batch(w, idx, dtype=np.float64))
        _check(err, scale, (n, l, b))
        if b > 1:       # and the same bits one by one
            assert np.array_equal(np.stack([e.forward(x) for x in idx]), got)


def test_reference_goldens_all_checkpoints(engines, golden, repo):
    """The 20 test MSAs x 5 checkpoints against the reference's own outputs."""
    g = golden("e2e_testdata.npz")
    worst = 0.0
    for ck in ("pf", "pf_base", "pf_indel", "pf_cherry", "pf_selreg"):
        e = engines(ck)
        for f in sorted(glob.glob(os.path.join(repo, "data/testdata/msas/*.fa"))):
            idx, _ = load_alignment(f)
            err = np.abs(e.forward(idx) - g[f"{ck}/{os.path.basename(f)[:-3]}"]).max()
            worst = max(worst, err)
            assert err <= TOL, (ck, f, err)
    _ERRORS["reference test MSAs, 20 x 5 checkpoints (worst)"] = {"max_abs_err": float(worst), "max_abs_ref": None}
    print(f"worst max-abs error over 100 reference outputs: {worst:.3e}")


def test_config_goldens(engines, golden):
    g = golden("configs.npz")
    e = engines("pf")
    got = e.forward(g["c2_idx"])
    err2, _ = _record("config 2: 20x200 x3, pf", got, g["c2_dist"])
    assert err2 <= TOL
    got3 = e.forward(g["c3_idx"])                      # headline shape 60 x 500
    err, _ = _record("config 3: 60x500, pf (headline)", got3, g["c3_dist"])
    assert err <= TOL


def test_config_goldens_more_headline_shape(engines, golden):
    """Three further 60 x 500 alignments (other seeds, pf.ckpt) and a gapped 60 x 500 one (pf_indel.ckpt),
    all against the reference's own outputs (oracle/gen_golden.py --only configs_more): absolute 1e-4."""
    g = golden("configs_more.npz")
    got = engines("pf").forward(g["c3b_idx"])
    for k in range(g["c3b_idx"].shape[0]):
        err, _ = _record(f"60x500 seed 31 #{k}, pf", got[k], g["c3b_dist"][k])
        assert err <= TOL
    gotg = engines("pf_indel").forward(g["c3g_idx"])
    assert (g["c3g_idx"] == 21).any(), "the gapped golden must contain gaps"
    err, _ = _record("60x500 gapped, pf_indel", gotg, g["c3g_dist"])
    assert err <= TOL


def test_model_surface_one_hot_input_and_squeeze(weights, golden):
    """The reference's call surface on the GPU path (model.py:166-187): one-hot float input [B, 22, L, N]
    as infer_alns.py:112 builds it, `torch.squeeze` semantics of the output ([P] for B = 1, [B, P] for a
    batch, 0-dim for two sequences) - values against the reference's own outputs (batch_small.npz)."""
    from phyloformer_amd import fasta
    from phyloformer_amd.model import Phyloformer
    g = golden("batch_small.npz")
    m = Phyloformer(weights("pf"), device=0)
    try:
        onehot = np.stack([fasta.one_hot(a) for a in g["idx"]]).astype(np.float32)      # [2, 22, L, N]
        assert onehot.shape == (2, 22, 40, 6)
        yb = m(onehot)
        assert yb.shape == g["dist"].shape == (2, 15)
        err, _ = _record("model(x) one-hot batch 2 x (6x40)", yb, g["dist"],
                         f64=O.forward_batch(weights("pf").tensors, g["idx"], dtype=np.float64))
        assert err <= TOL
        y1 = m(onehot[:1])
        assert y1.shape == (15,) and np.abs(y1 - g["dist"][0]).max() <= TOL          # B = 1 squeezes to [P]
        y2 = m(fasta.one_hot(g["idx_n2"]).astype(np.float32)[None])
        assert y2.shape == () == g["dist_n2"].shape                                   # N = 2: 0-dim
        err2, _ = _record("model(x) N = 2 (0-dim)", y2, g["dist_n2"])
        assert err2 <= TOL
        # index input keeps its batch axis, as documented
        assert m(g["idx"]).shape == (2, 15)
    finally:
        m.close()


def test_config_goldens_big(engines, golden, repo):
    path = os.path.join(repo, "tests/golden/configs_big.npz")
    if not os.path.exists(path):
        pytest.skip("configs_big.npz not generated")
    g = np.load(path)
    err4, _ = _record("config 4: 60x2000, pf", engines("pf").forward(g["c4_idx"]), g["c4_dist"])
    err5, _ = _record("config 5: 200x500 gapped, pf_indel", engines("pf_indel").forward(g["c5_idx"]), g["c5_dist"])
    assert err4 <= TOL and err5 <= TOL


def test_batch_invariance_and_determinism(engines, golden):
    g = golden("configs.npz")
    e = engines("pf")
    a = g["c2_idx"]
    batch = e.forward(a)
    single = np.stack([e.forward(x) for x in a])
    # no atomics, and no launch parameter that touches the order of a sum depends on the batch size:
    # an alignment gets the same bits alone, in a batch of 3, or in a batch of 40 cut into workspace chunks
    assert np.array_equal(batch, single)
    assert np.array_equal(e.forward(a), batch)
    big = e.forward(np.concatenate([a] * 14)[:40])
    assert np.array_equal(big[:3], batch) and np.array_equal(big[39], batch[39 % 3])
    g3 = g["c3_idx"]
    assert np.array_equal(e.forward(np.concatenate([g3, g3, g3]))[2], e.forward(g3)[0])


def test_column_statistics_by_groups_or_by_runs_same_bits(weights, golden):
    """k_colstats walks a pair group per block, or - when groups alone would leave the chip idle, e.g. a lone
    alignment - one run of the group per block with k_colfin folding the runs (pf_lib.hip colstats_plan).
    Both realise the same summation tree: identical bits, forced either way or chosen by batch size, also
    where the last group is short and where a group is a single run."""
    from phyloformer_amd.engine import Engine
    g = golden("configs.npz")
    cases = [g["c2_idx"], g["c3_idx"], simulate_batch(2, 23, 45, seed=5), simulate_batch(1, 5, 33, seed=6),
             simulate_batch(1, 75, 40, seed=7)]
    for idx in cases:
        out = {}
        for fine in (0, 1, -1):
            with Engine(weights("pf"), 0) as e:
                e.set_option("precise", 0)
                e.set_option("colstats_fine", fine)
                out[fine] = e.forward(idx)
        assert np.array_equal(out[0], out[1]) and np.array_equal(out[0], out[-1])


def test_two_stream_schedule_same_bits(weights, golden):
    """A batch of >= 2 alignments runs as two half-batches on two streams by default (option two_streams);
    the halves are independent forwards, so the distances are those of the one-stream schedule bit for bit -
    for even and odd batches, and with back-to-back asynchronous forwards sharing the two workspaces."""
    from phyloformer_amd.engine import Engine
    g = golden("configs.npz")
    for idx in (g["c2_idx"], np.concatenate([g["c2_idx"], g["c2_idx"][:2]]), simulate_batch(2, 31, 90, seed=9)):
        out = {}
        for ts in (0, 1):
            with Engine(weights("pf"), 0) as e:
                e.set_option("two_streams", ts)
                out[ts] = e.forward(idx)
                if ts:
                    B, n, l = idx.shape
                    P = n * (n - 1) // 2
                    d_idx, d_out = e.malloc(idx.nbytes), e.malloc(B * P * 4)
                    e.h2d(d_idx, np.ascontiguousarray(idx))
                    for _ in range(3):
                        e.forward_device(d_idx, B, n, l, d_out)
                    again = np.empty((B, P), np.float32)
                    e.d2h(again, d_out)
                    assert np.array_equal(again, out[1])
        assert np.array_equal(out[0], out[1])


def test_permutation_equivariance(engines):
    e = engines("pf", precise=0)        # the default kernels (option pinned: the test sweeps their tilings)
    idx = simulate_batch(1, 9, 70, seed=21)[0]
    base = e.forward(idx)
    rng = np.random.default_rng(0)
    perm = rng.permutation(70)
    assert np.abs(e.forward(idx[:, perm]) - base).max() <= 2e-5     # sites are exchangeable
    sp = rng.permutation(9)
    from phyloformer_amd.phylip import vec_to_matrix
    dm, dmp = vec_to_matrix(base, 9), vec_to_matrix(e.forward(idx[sp]), 9)
    assert np.abs(dmp - dm[np.ix_(sp, sp)]).max() <= 2e-5          # and so are sequences


def test_errors_mirror_reference(engines):
    e = engines("pf")
    with pytest.raises(ValueError, match="n_seqs must be smaller or equal to 200"):
        e.forward(np.zeros((201, 8), np.uint8))                      # model.py:24-28
    with pytest.raises(ValueError, match="outside 0..21"):
        e.forward(np.full((3, 8), 22, np.uint8))
    # one sequence = no pair: the reference's forward fails in attention.py:193 with this RuntimeError
    # (tests/golden/cli_bad_entry.json, from the real CLI); the C ABI itself answers PF_EINVAL for N < 2
    with pytest.raises(RuntimeError, match=r"cannot reshape tensor of 0 elements into shape \[1, -1, 0, 64\]"):
        e.forward(np.zeros((1, 8), np.uint8))
    out = np.zeros(4, np.float32)
    assert e._lib.pf_forward(e._h, np.zeros((1, 8), np.uint8).ctypes.data, 1, 1, 8, out.ctypes.data) == -1
    e.set_option("max_seqs", 0)                                      # opt-in: lift the cap
    try:
        assert e.forward(np.zeros((201, 2), np.uint8)).shape == (201 * 200 // 2,)
    finally:
        e.set_option("max_seqs", 200)


def test_full_range_shard_equals_forward(engines, golden):
    g = golden("configs.npz")
    e = engines("pf")
    a = g["c2_idx"][:1]
    assert np.array_equal(e.forward_sharded(a, 0, 200, 200), e.forward(a))


def test_table_embedding_equals_mfma_embedding(engines, golden):
    """Block 0's row statistics come from a host-built residue-pair table (k_embed); the MFMA
    formulation it replaced (k_main<MODE_FIRST>, option "embed_mfma") must give the same taps and
    the same distances."""
    e = engines("pf_indel", precise=0)      # the default kernels' block 0 (option pinned)
    rng = np.random.default_rng(12)
    idx = rng.integers(0, 22, (2, 9, 75)).astype(np.uint8)          # all 22 symbols incl. X and gap
    taps = {}
    for mode in (0, 1):
        e.set_option("embed_mfma", mode)
        e.set_option("debug_keep", 1)
        d = e.forward(idx)
        taps[mode] = (d, e.debug_read("x0"), e.debug_read("srow0"))
        e.set_option("debug_keep", 0)
    e.set_option("embed_mfma", 0)
    assert np.array_equal(taps[0][1], taps[1][1])                    # x0: the same fp32 sums
    s0, s1 = taps[0][2].reshape(-1, 72), taps[1][2].reshape(-1, 72)
    assert np.abs(s0 - s1).max() <= 2e-5 * np.abs(s1).max()
    # distances: random residues are an ill-conditioned input (see test_tiny_taps...), so compare on the
    # scale of the output there and absolutely on a simulated alignment
    assert np.abs(taps[0][0] - taps[1][0]).max() <= 5e-5 * max(1.0, np.abs(taps[1][0]).max())
    g = golden("configs.npz")
    e.set_option("embed_mfma", 1)
    d1 = e.forward(g["c2_idx"])
    e.set_option("embed_mfma", 0)
    d0 = e.forward(g["c2_idx"])
    assert np.abs(d0 - d1).max() <= 2e-5


def test_workspace_chunking_does_not_change_results(engines, golden):
    """A batch that does not fit the workspace budget is cut into chunks (pf_set_option "ws_limit_mb");
    the chunked run returns the same bits."""
    e = engines("pf")
    a = np.concatenate([golden("configs.npz")["c2_idx"]] * 6)[:17]          # 17 alignments of 20 x 200
    whole = e.forward(a)
    e.set_option("ws_limit_mb", 64)                                           # ~5 alignments per chunk
    try:
        chunked = e.forward(a)
    finally:
        e.set_option("ws_limit_mb", 24576)
    assert np.array_equal(whole, chunked)


def test_alternative_kernel_paths_agree(weights, golden):
    """Block 0 forms x0 = T[a_i] + T[a_j] on the fly (default) or reads the materialised copy (option
    materialize_x0): the same fp32 sums, so bit-identical distances.  (k_main2 and k_colstats2, the measured
    alternatives of round 2, were retired in round 3: profiles/r03_energy_*.)"""
    from phyloformer_amd.engine import Engine
    a = golden("configs.npz")["c2_idx"]
    out = {}
    for opt in (None, "materialize_x0"):
        with Engine(weights("pf"), 0) as e:
            if opt:
                e.set_option(opt, 1)
            out[opt] = e.forward(a)
    assert np.array_equal(out[None], out["materialize_x0"])
    assert np.abs(out[None] - golden("configs.npz")["c2_dist"]).max() <= TOL


def test_flat_tiling_against_row_tiling(weights, golden):
    """k_main cuts an alignment's P x L tokens into 32-token tiles that may cover the end of one pair row and the
    start of the next (flat tiling; chosen when L >= 32 and L % 32 != 0), instead of giving every row its own
    ragged last tile.  PF_ROW_TILES=1 forces the row tiling: same distances up to the grouping of the per-row
    partial sums (<= 2e-5 of the largest distance), for row lengths just above / below tile multiples, rows
    shorter than two tiles, a one-row alignment, and batches whose alignments start at any token offset."""
    from phyloformer_amd.engine import Engine
    cases = [simulate_batch(3, 7, 33, seed=41), simulate_batch(2, 6, 63, seed=42), simulate_batch(1, 9, 65, seed=43),
             simulate_batch(2, 2, 45, seed=44), simulate_batch(1, 14, 97, seed=45, gaps=True),
             golden("configs.npz")["c2_idx"], simulate_batch(5, 4, 40, seed=46)]
    for idx in cases:
        out = {}
        for row_tiles in (0, 1):
            # (the host picks the tiling from the shape - flat only where it pays, e.g. not at L = 63 -: force both)
            os.environ["PF_ROW_TILES" if row_tiles else "PF_FLAT_TILES"] = "1"
            try:
                with Engine(weights("pf"), 0) as e:
                    e.set_option("precise", 0)
                    out[row_tiles] = e.forward(idx)
                    if not row_tiles:          # flat: an alignment's bits do not depend on its place in the batch
                        assert np.array_equal(np.stack([e.forward(x) for x in idx]), out[0])
                        assert np.array_equal(e.forward_shards_emulated(idx, 1), out[0])
            finally:
                os.environ.pop("PF_ROW_TILES", None)
                os.environ.pop("PF_FLAT_TILES", None)
        err = np.abs(out[0] - out[1]).max()
        assert err <= 2e-5 * max(1.0, float(np.abs(out[1]).max())), (idx.shape, err)


def test_shape_sweep_against_oracle(engines, weights):
    """Forty seeded (n_seqs, n_sites, batch) shapes - row lengths on both sides of every tile multiple up to 200, one
    to 105 pairs, i.e. every way a 32-token tile can meet a row end or an alignment end in either tiling - against
    the oracle (absolute 1e-4), with the same bits one alignment at a time.  Default kernels on every shape: this is
    their tiling sweep (the product's routing of small shapes is tests/test_gpu_precise.py's soak)."""
    e = engines("pf", precise=0)
    w = weights("pf").tensors
    rng = np.random.default_rng(2024)
    shapes = [(int(rng.integers(2, 16)), int(rng.integers(1, 201)), int(rng.integers(1, 4))) for _ in range(32)]
    shapes += [(3, 32, 2), (3, 64, 1), (15, 96, 2), (2, 33, 3), (11, 31, 1), (4, 160, 2), (5, 161, 1), (6, 159, 2)]
    worst = 0.0
    for i, (n, l, b) in enumerate(shapes):
        idx = simulate_batch(b, n, l, seed=100 + i, gaps=(i % 4 == 0))
        got = e.forward(idx)
        want = O.forward_batch(w, idx)
        err = float(np.abs(got - want).max())
        worst = max(worst, err)
        assert err <= TOL, (n, l, b, err)
        if b > 1:
            assert np.array_equal(np.stack([e.forward(x) for x in idx]), got), (n, l, b)
    _ERRORS["shape sweep: 40 seeded shapes, 2-15 sequences x 1-200 sites (worst)"] = {"max_abs_err": worst, "max_abs_ref": None}
    print(f"shape sweep: worst max-abs error {worst:.3e}")


def test_device_entry_points_survive_out_of_alphabet_bytes(weights, golden):
    """VERDICT r03 / next 4.  pf_forward refuses a residue byte > 21 on the host (ValueError; the reference raises
    KeyError, data.py:25-26).  The device entry points take buffers the library never saw: every table lookup clamps
    (255 reads row 21, never 60 KB past a 5.6 KB table), k_embed raises a sticky flag and the next synchronising call
    (pf_synchronize / pf_memcpy_d2h) reports PF_EINVAL once.  Never a fault; the handle stays usable."""
    from phyloformer_amd.engine import Engine
    a = golden("configs.npz")["c2_idx"][:2].copy()        # 2 x (20 x 200)
    B, N, L = a.shape
    P = N * (N - 1) // 2
    bad = a.copy()
    bad[0, 3, 17] = 255
    bad[1, 5, 150:153] = 22
    as_gap = np.minimum(bad, 21)
    with Engine(weights("pf"), 0) as e:
        want = e.forward(as_gap)
        with pytest.raises(ValueError, match="outside 0..21"):
            e.forward(bad)
        d_idx, d_out = e.malloc(bad.nbytes), e.malloc(B * P * 4)
        out = np.empty((B, P), np.float32)
        for opts in ({}, {"materialize_x0": 1}, {"two_streams": 0}):
            for k, v in opts.items():
                e.set_option(k, v)
            e.h2d(d_idx, bad)
            e.forward_device(d_idx, B, N, L, d_out)
            with pytest.raises(ValueError, match="outside 0..21"):
                e.synchronize()
            e.d2h(out, d_out)                              # the flag was consumed: this one succeeds
            assert np.isfinite(out).all() and np.array_equal(out, want), opts
            for k in opts:
                e.set_option(k, 1 if k == "two_streams" else 0)
        e.h2d(d_idx, a)                                    # valid input afterwards: no stale flag, the usual bits
        e.forward_device(d_idx, B, N, L, d_out)
        e.synchronize()
        e.d2h(out, d_out)
        assert np.array_equal(out, e.forward(a))
        e.set_option("embed_mfma", 1)                      # cross-check path (no k_embed): clamped, finite, no flag
        e.h2d(d_idx, bad)
        e.forward_device(d_idx, B, N, L, d_out)
        e.synchronize()
        e.d2h(out, d_out)
        # (k_main<FIRST> computes block 0's statistics by split-fp16 MFMA instead of the fp64-built table: fp32 noise
        # in distribution; this input carries a residue that never co-occurs with the others in training)
        assert np.isfinite(out).all() and np.abs(out - want).max() <= 1e-3
        e.free(d_idx)
        e.free(d_out)


@pytest.mark.parametrize("ck,n,l,gaps", [("pf", 60, 500, False), ("pf_indel", 200, 500, True), ("pf", 200, 2000, False)])
def test_full_size_properties(engines, ck, n, l, gaps):
    """BASELINE's full sizes (configs[2] 60 x 500, configs[4] gapped 200 x 500) and the largest alignment the reference's
    cap admits at configs[3]'s length (200 x 2000: 39.8 M tokens, a 10 GB residual stream), where the oracle takes minutes, through
    properties that do not depend on the size: sites are exchangeable (the distance is a mean over sites of a function
    that is permutation-equivariant along them, model.py:166-187), sequences are equivariant (permuting them permutes
    the distance matrix), an alignment's bits do not depend on its neighbours in the batch, and eight emulated site
    shards (ragged: 500 = 7 x 63 + 59) give the unsharded distances."""
    from phyloformer_amd.phylip import vec_to_matrix
    e = engines(ck)
    # (the Python simulator is slow for 400,000 residues: the largest case tiles a 200 x 250 alignment along the sites)
    idx = simulate_batch(1, n, min(l, 250) if n * l > 200_000 else l, seed=2025 + n, gaps=gaps)[0]
    idx = np.ascontiguousarray(np.tile(idx, (1, l // idx.shape[1])))
    assert idx.shape == (n, l)
    base = e.forward(idx)
    scale = max(1.0, float(np.abs(base).max()))
    assert np.isfinite(base).all() and (base > 0).all()
    rng = np.random.default_rng(n)
    sites = rng.permutation(l)
    assert float(np.abs(e.forward(idx[:, sites]) - base).max()) <= 2e-5 * scale
    seqs = rng.permutation(n)
    dm, dmp = vec_to_matrix(base, n), vec_to_matrix(e.forward(idx[seqs]), n)
    assert float(np.abs(dmp - dm[np.ix_(seqs, seqs)]).max()) <= 2e-5 * scale
    if n <= 60:
        other = simulate_batch(2, n, l, seed=7, gaps=gaps)
        both = e.forward(np.stack([other[0], idx, other[1]]))
        assert np.array_equal(both[1], base)                               # batch invariance, bitwise
    assert float(np.abs(e.forward_shards_emulated(idx, 8) - base).max()) <= 2e-5 * scale


def test_colstats_ring_prefetch_is_bit_identical_to_register_prefetch(engines, golden):
    """Round 6: k_colstats<false, RING> moves the token rows and q' of the next pairs global -> LDS (global_load_lds
    into a per-wave ring, counted s_waitcnt vmcnt) instead of through registers; the arithmetic and its order are the
    register variant's (option colstats_ring = 0), so every bit must agree: batched and alone (groups / runs), ragged
    site chunks, short groups, a shape whose last chunk holds one site."""
    e = engines("pf", precise=0)
    g = golden("configs.npz")
    rng = np.random.default_rng(77)
    cases = [g["c3_idx"][:3], g["c3_idx"][:1], g["c2_idx"], rng.integers(0, 22, (2, 9, 65)).astype(np.uint8),
             rng.integers(0, 20, (1, 33, 40)).astype(np.uint8), rng.integers(0, 20, (5, 7, 97)).astype(np.uint8)]
    try:
        for idx in cases:
            e.set_option("colstats_ring", 1)
            ring = e.forward(idx)
            e.set_option("colstats_ring", 0)
            regs = e.forward(idx)
            assert np.isfinite(ring).all() and np.array_equal(ring, regs), idx.shape
    finally:
        e.set_option("colstats_ring", 1)
