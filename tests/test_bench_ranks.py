"""bench.py's rank logic (site slicing, agreed fallback, JSON assembly) at world size 2 over the TCP
rendezvous with a fake engine - the N > 1 launch path the driver runs on an 8-GPU node, minus the GPU."""
import io
import json
import multiprocessing as mp
import os
import sys
import uuid

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class FakeEngine:
    """Engine interface used by bench.run; records what it is asked to do."""

    def __init__(self, rank, log, fail_comm_on=None):
        self.rank, self.log, self.fail_comm_on = rank, log, fail_comm_on
        self.bufs, self.comm = {}, False
        self.calls = 0

    def set_option(self, k, v): self.log.append(("opt", k, v))
    def unique_id(self): return bytes(range(1, 129))

    def comm_init(self, uid, rank, world):
        if self.fail_comm_on == rank:
            raise RuntimeError("ncclCommInitRank failed: unhandled system error")
        assert uid == bytes(range(1, 129))
        self.comm = True
        self.log.append(("comm_init", rank, world))

    def comm_info(self): return {"library": "/opt/rocm/lib/librccl.so.1", "version": 22707}
    def comm_destroy(self): self.comm = False; self.log.append(("comm_destroy",))
    def malloc(self, n): self.bufs[len(self.bufs) + 1] = n; return len(self.bufs)
    def free(self, p): self.bufs.pop(p)
    def h2d(self, d, a): self.log.append(("h2d", a.shape))
    def d2h(self, out, d): out[...] = 0.25
    def synchronize(self): pass
    def profile_reset(self): pass
    def profile_get(self, k): return (6 * self.calls, 4.0 * 6 * self.calls)
    def device_info(self): return {"name": "fake gfx950", "cu_count": 256, "hbm_bytes": 1 << 38}
    def close(self): self.log.append(("close",))

    def forward_sharded_device(self, d_idx, B, N, lo, hi, L, d_out):
        assert self.comm, "site-sharded step without a communicator"
        self.calls += 1
        self.log.append(("sharded", B, N, lo, hi, L))

    def forward_device(self, d_idx, B, N, L, d_out):
        assert not self.comm, "plain forward on a handle that still carries a communicator"
        self.calls += 1
        self.log.append(("plain", B, N, L))

    def forward_sharded(self, idx, lo, hi, L):
        assert self.comm
        return np.full((idx.shape[0], idx.shape[1] * (idx.shape[1] - 1) // 2), 0.25, np.float32)

    def forward(self, idx):
        assert not self.comm
        return np.full((idx.shape[0], idx.shape[1] * (idx.shape[1] - 1) // 2), 0.25, np.float32)


class _W:
    n_blocks = 6


def _worker(rank, world, key, tmp, fail_comm_on, q):
    sys.path.insert(0, REPO)
    import bench
    from phyloformer_amd.rendezvous import TcpGroup
    args = bench.parse_args(["--gpus", str(world), "--steps", "3", "--warmup", "1", "--batch", "2", "--n-seqs", "6",
                             "--n-sites", "45", "--no-power"])
    log, out = [], io.StringIO()
    with TcpGroup(rank, world, key=key, directory=tmp, timeout=30) as g:
        bench.run(args, rank, world, rank, g, lambda dev: FakeEngine(rank, log, fail_comm_on), _W(), out=out)
    q.put((rank, log, out.getvalue()))


@pytest.mark.parametrize("fail_comm_on", [None, 1])
def test_bench_rank_logic_world2(fail_comm_on, tmp_path):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    key = "pf_bench_" + uuid.uuid4().hex
    procs = [ctx.Process(target=_worker, args=(r, 2, key, str(tmp_path), fail_comm_on, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict((r, (log, text)) for r, log, text in (q.get(timeout=120) for _ in range(2)))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    line = json.loads(res[0][1])
    assert res[1][1] == "", "only rank 0 prints"
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert line["value"] > 0 and line["value_host_buffers"] > 0 and line["cpu_baseline"] is None
    assert abs(line["value"] * line["ms_per_step"] * 1e-3 - line["config"]["global_batch"]) < 1e-2 * line["config"]["global_batch"]
    if fail_comm_on is None:
        # sites 0..22 on rank 0, 23..44 on rank 1 (ceil split), global batch = batch x world
        for rank, (lo, hi) in enumerate([(0, 23), (23, 45)]):
            steps = [e for e in res[rank][0] if e[0] == "sharded"]
            # warm-up + 3 timed steps on two streams, then 1 + 3 with the batch on one stream (roofline region)
            assert len(steps) == 8 and all(e == ("sharded", 4, 6, lo, hi, 45) for e in steps)
            opts = [e[1:] for e in res[rank][0] if e[0] == "opt" and e[1] in ("two_streams", "overlap")]
            assert opts == [("two_streams", 1), ("overlap", 1), ("two_streams", 0), ("overlap", 0),
                            ("two_streams", 1), ("overlap", 1)]
            assert ("h2d", (4, 6, hi - lo)) in res[rank][0]
        assert line["config"]["parallelism"] == "sites-sharded x2" and line["config"]["global_batch"] == 4
        assert line["config"]["rccl"]["library"].endswith("librccl.so.1")
        assert line["roofline"]["launches"] == 6 * 8 and line["roofline"]["traffic_source"].startswith("profiles/")
        assert line["value_one_stream"] > 0 and "one stream" in line["roofline"]["schedule"]
    else:
        # rank 1's communicator failed: BOTH ranks destroy theirs and shard whole alignments instead
        for rank in (0, 1):
            log = res[rank][0]
            assert ("comm_destroy",) in log
            assert not [e for e in log if e[0] == "sharded"]
            assert len([e for e in log if e == ("plain", 2, 6, 45)]) == 8
        assert line["config"]["parallelism"] == "alignments-sharded x2" and line["config"]["global_batch"] == 4
        assert "RCCL init failed" in line["config"]["note"]
