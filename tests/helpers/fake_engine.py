"""Stand-ins for the Engine used by the CPU tests of bench.py's rank logic (no GPU, no native library)."""
import os

import numpy as np

_GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "golden")
_ANSWERS = {(20, 200): ("configs.npz", "c2_dist"), (60, 500): ("configs.npz", "c3_dist"),
            (60, 2000): ("configs_big.npz", "c4_dist")}


def _answer(B, N, L_total):
    """What a correct engine returns for the committed goldens (the parity leg of bench.py feeds exactly those
    alignments, repeated): the reference distances themselves; 0.25 for any other shape."""
    hit = _ANSWERS.get((N, L_total))
    if hit is None:
        return np.full((B, N * (N - 1) // 2), 0.25, np.float32)
    dist = np.load(os.path.join(_GOLDEN_DIR, hit[0]))[hit[1]]
    return np.ascontiguousarray(dist[np.arange(B) % dist.shape[0]])


class FakeEngine:
    """Engine interface used by bench.run; records what it is asked to do."""

    def __init__(self, rank, log, fail_comm_on=None):
        self.rank, self.log, self.fail_comm_on = rank, log, fail_comm_on
        self.bufs, self.comm = {}, False
        self.calls = 0
        self.last = None

    def set_option(self, k, v): self.log.append(("opt", k, v))
    def unique_id(self): return bytes(range(256))[::-1]

    def comm_init(self, uid, rank, world):
        if self.fail_comm_on == rank:
            raise RuntimeError("ncclCommInitRank failed: unhandled system error")
        assert uid == bytes(range(256))[::-1]
        self.comm = True
        self.log.append(("comm_init", rank, world))

    def comm_info(self): return {"library": "/opt/rocm/lib/librccl.so.1", "version": 22707}
    def comm_destroy(self): self.comm = False; self.log.append(("comm_destroy",))
    def malloc(self, n): self.bufs[len(self.bufs) + 1] = n; return len(self.bufs)
    def free(self, p): self.bufs.pop(p)
    def h2d(self, d, a): self.log.append(("h2d", a.shape))
    def d2h(self, out, d):
        out[...] = self.last if (self.last is not None and self.last.shape == out.shape) else 0.25
        if os.environ.get("PF_FAKE_PARITY_ERROR") == str(self.rank) and self.last is not None and self.last.shape == out.shape:
            out[0, 0] += 1e-3           # a rank whose result is off: bench must say so and exit non-zero
    def synchronize(self): pass
    def profile_reset(self): pass
    def collective_count(self): return 14 * self.calls if self.comm else 0
    def profile_get(self, k): return (6 * self.calls, 4.0 * 6 * self.calls)
    def device_info(self): return {"name": "fake gfx950", "cu_count": 256, "hbm_bytes": 1 << 38}
    def close(self):
        self.log.append(("close",))
        if getattr(self, "log_path", None):
            import json
            with open(self.log_path, "w") as fh:
                json.dump([list(e) for e in self.log], fh)

    def forward_sharded_device(self, d_idx, B, N, lo, hi, L, d_out):
        assert self.comm, "site-sharded step without a communicator"
        self.calls += 1
        self.last = _answer(B, N, L)
        self.log.append(("sharded", B, N, lo, hi, L))

    def forward_device(self, d_idx, B, N, L, d_out):
        assert not self.comm, "plain forward on a handle that still carries a communicator"
        self.calls += 1
        self.last = _answer(B, N, L)
        self.log.append(("plain", B, N, L))

    def forward_sharded(self, idx, lo, hi, L):
        assert self.comm
        return np.full((idx.shape[0], idx.shape[1] * (idx.shape[1] - 1) // 2), 0.25, np.float32)

    def forward(self, idx):
        assert not self.comm
        return np.full((idx.shape[0], idx.shape[1] * (idx.shape[1] - 1) // 2), 0.25, np.float32)


class FakeWeights:
    n_blocks = 6


def bench_weights():
    return FakeWeights()


def make_for_bench(device):
    """Factory named by PF_BENCH_ENGINE_FACTORY=tests.helpers.fake_engine:make_for_bench (self-launch test): the
    rank comes from the launcher's environment, the log goes to $PF_FAKE_LOG_DIR/rank<r>.log at close()."""
    rank = int(os.environ.get("RANK", "0"))
    fail = os.environ.get("PF_FAKE_FAIL_COMM_ON")
    eng = FakeEngine(rank, [], None if fail in (None, "") else int(fail))
    eng.log_path = os.path.join(os.environ["PF_FAKE_LOG_DIR"], f"rank{rank}.log") if os.environ.get("PF_FAKE_LOG_DIR") else None
    return eng


def make_dying_on_rank1(device):
    if os.environ.get("RANK") == "1":
        os._exit(7)
    return make_for_bench(device)


def make_hanging(device):
    import time
    time.sleep(600)


def make_hanging_on_rung1(device):
    """A collective that never completes on the two-stream rung (a rank spins in its first site-sharded step);
    the one-stream rung works."""
    eng = make_for_bench(device)
    if os.environ.get("PF_BENCH_RUNG") == "1":
        def stuck(*a, **k):
            import time
            time.sleep(600)
        eng.forward_sharded_device = stuck
    return eng


def make_failing_until_rung3(device):
    """Rung 1 hangs, on rung 2 rank 1 dies; only the alignment-sharded rung (no collective) gets through."""
    rung = os.environ.get("PF_BENCH_RUNG")
    if rung == "2" and os.environ.get("RANK") == "1":
        os._exit(9)
    return make_hanging_on_rung1(device)


def make_slow(device):
    """A healthy but slow GPU: launches return at once, every synchronisation then takes 0.15 s per step queued since
    the last one - a timed region of K steps is ONE long silence, as on the real engine."""
    import time
    eng = make_for_bench(device)
    queued = [0]
    fs, fp, sync = eng.forward_sharded_device, eng.forward_device, eng.synchronize

    def fsd(*a, **k):
        queued[0] += 1
        return fs(*a, **k)

    def fd(*a, **k):
        queued[0] += 1
        return fp(*a, **k)

    def synchronize():
        time.sleep(0.15 * queued[0])
        queued[0] = 0
        return sync()

    def d2h(out, d, _orig=eng.d2h):
        synchronize()
        return _orig(out, d)
    eng.forward_sharded_device, eng.forward_device, eng.synchronize, eng.d2h = fsd, fd, synchronize, d2h
    return eng
