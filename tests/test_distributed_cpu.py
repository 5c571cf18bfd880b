"""N > 1 on CPU: two gloo ranks run the site-sharded algorithm with real collectives.

The device path cannot run here (no GPU), so the ranks evaluate the oracle's per-rank
form (oracle/pf_oracle.py::forward_rank) — the same collective schedule the native
library issues with RCCL: one fused [P, 72] all-reduce per block plus one [P] at the end —
and exercise the host plumbing of phyloformer_amd/dist.py (site ranges, unique-id broadcast).
"""
import os
import socket
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, L, q, dtype="float32"):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from oracle import pf_oracle as O
    from phyloformer_amd import dist as pfdist
    from phyloformer_amd.msa_sim import simulate_batch
    from phyloformer_amd.weights import load_weights

    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = []

    def allreduce(a):
        t = torch.from_numpy(a.copy())
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        calls.append(tuple(a.shape))
        return t.numpy()

    w = load_weights(os.path.join(REPO, "models", "pf.ckpt"))
    idx = simulate_batch(1, 7, L, seed=42)[0]            # every rank builds the same alignment
    local, lo, hi = pfdist.shard_sites(idx, world, rank)
    d = O.forward_rank(w.tensors, local, L, allreduce, dtype=np.dtype(dtype))
    calls.append(str(d.dtype))
    uid = pfdist.broadcast_bytes(bytes(range(128)) if rank == 0 else None, 128, src=0)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, lo, hi, d, calls, uid))


@pytest.mark.parametrize("world,L", [(2, 45), (3, 2)])
def test_site_sharded_ranks_match_unsharded(world, L, weights):
    import torch.multiprocessing as mp
    from oracle import pf_oracle as O
    from phyloformer_amd.msa_sim import simulate_batch
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, L, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = O.forward(weights("pf").tensors, simulate_batch(1, 7, L, seed=42)[0])
    covered = []
    for rank, lo, hi, d, calls, uid in res:
        assert np.abs(d - want).max() <= 1e-5               # every rank ends with the full result
        assert calls == [(21, 72)] * 6 + [(21,), "float32"]  # 6 fused statistics + 1 final all-reduce
        assert uid == bytes(range(128))
        covered += list(range(lo, hi))
    assert covered == list(range(L))                        # ranges tile the sites (empty rank allowed)


def test_float64_ranks_match_the_float64_forward(weights):
    """The float64 path's collective schedule (csrc/pf_precise_host.hip.h: n_blocks double all-reduces of [P, 72] and one
    of [P], never cut into halves) at world 2 with real gloo all-reduces of doubles: 7 x 45 sites split 23 + 22, every
    rank ends with the unsharded float64 forward to 1e-12."""
    import torch.multiprocessing as mp
    from oracle import pf_oracle as O
    from phyloformer_amd.msa_sim import simulate_batch
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 45, q, "float64")) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = O.forward(weights("pf").tensors, simulate_batch(1, 7, 45, seed=42)[0], dtype=np.float64)
    for rank, lo, hi, d, calls, uid in res:
        assert d.dtype == np.float64 and np.abs(d - want).max() <= 1e-12 * max(1.0, np.abs(want).max())
        assert calls == [(21, 72)] * 6 + [(21,), "float64"]
