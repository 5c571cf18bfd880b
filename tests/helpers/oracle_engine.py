"""An Engine stand-in with REAL numerics for CPU tests of the CLI's multi-rank plumbing (no GPU, no native library):
every forward goes through oracle/pf_oracle.py, and a site-sharded forward really exchanges its row statistics with
the other ranks - over a TcpGroup of its own, summed in rank order.  tests/ may use the oracle; the product never does.

    PF_CLI_ENGINE_FACTORY=helpers.oracle_engine:make

``PF_ORACLE_SHARDS=k`` makes the plain ``forward`` evaluate the k-shard algorithm in one process (shard sums in
shard order): the association a k-rank run has, so that its outputs can be compared byte for byte."""
import os

import numpy as np

from oracle import pf_oracle as O
from phyloformer_amd.rendezvous import TcpGroup


class OracleEngine:
    def __init__(self, weights, device):
        self.w = weights.tensors
        self.device, self.rank, self.world, self.comm = device, 0, 1, None
        self.ncoll = 0

    # -- Engine interface used by infer_alns.py / scheduler.py
    def set_option(self, k, v): pass
    def close(self):
        if self.comm is not None:
            self.comm.close()
            self.comm = None

    def unique_id(self): return os.urandom(256)
    def comm_info(self): return {"library": "oracle_engine (TcpGroup)", "version": 0}
    def collective_count(self): return self.ncoll

    def comm_init(self, uid, rank, world):
        if os.environ.get("PF_FAKE_FAIL_COMM_ON") == str(rank):
            raise RuntimeError("ncclCommInitRank failed: unhandled system error (injected)")
        self.rank, self.world = rank, world
        if world > 1:
            # (a peer that failed before this point never joins: with the injected failure the others give up fast,
            # the way a real ncclCommInitRank reports a bootstrap error on every rank)
            self.comm = TcpGroup(rank, world, key="oracle_comm_" + bytes(uid)[:16].hex(),
                                 timeout=4 if os.environ.get("PF_FAKE_FAIL_COMM_ON") else
                                 600 if (os.environ.get("PF_FAKE_DIE_RANK") or os.environ.get("PF_FAKE_HANG_RANK")) else 60)

    def comm_destroy(self):
        self.close()
        self.rank, self.world = 0, 1

    def _allreduce(self, a):
        self.ncoll += 1
        if self.comm is None:
            return a
        parts = self.comm.gather(a.tobytes())
        total = None
        if self.rank == 0:
            total = np.frombuffer(parts[0], a.dtype).copy()
            for p in parts[1:]:
                total += np.frombuffer(p, a.dtype)           # rank order, like oracle.forward(shards=k)
            total = total.tobytes()
        return np.frombuffer(self.comm.broadcast(total), a.dtype).reshape(a.shape).copy()

    def forward(self, idx):
        idx = np.asarray(idx, np.uint8)
        shards = int(os.environ.get("PF_ORACLE_SHARDS", "1"))
        one = idx.ndim == 2
        out = np.stack([O.forward(self.w, a, shards=shards) for a in (idx[None] if one else idx)])
        return out[0] if one else out

    def forward_sharded(self, idx_local, lo, hi, L):
        if os.environ.get("PF_FAKE_DIE_RANK") == str(self.rank) and self.ncoll > 0:
            os._exit(7)                              # a rank lost mid-run (OOM kill, HIP error): its peers wait in a collective
        if os.environ.get("PF_FAKE_SLOW_S"):          # a long healthy launch (the stall watchdog must not take it for a hang)
            import time
            time.sleep(float(os.environ["PF_FAKE_SLOW_S"]))
        if os.environ.get("PF_FAKE_HANG_RANK") == str(self.rank) and self.ncoll > 0:
            import time
            time.sleep(3600)                         # a rank that stops making progress without dying
        idx_local = np.asarray(idx_local, np.uint8)
        assert idx_local.shape[-1] == hi - lo
        return np.stack([O.forward_rank(self.w, a, L, self._allreduce) for a in idx_local])


def make(weights, device):
    return OracleEngine(weights, device)
