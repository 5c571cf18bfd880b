"""A lone alignment, forwards back to back with device-resident indices: run under
`rocprofv3 --kernel-trace --stats` for true per-kernel durations (HIP events add ~3 us per launch).
    python tools/lone_profile.py [seqs] [sites] [forwards]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from phyloformer_amd.engine import Engine
from phyloformer_amd.weights import load_weights
from phyloformer_amd.msa_sim import simulate_batch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
l = int(sys.argv[2]) if len(sys.argv) > 2 else 200
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 300
repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
e = Engine(load_weights(os.path.join(repo, "models/pf.ckpt")), 0)
idx = np.ascontiguousarray(simulate_batch(1, n, l, seed=3))
P = n * (n - 1) // 2
d_idx = e.malloc(idx.nbytes); d_out = e.malloc(P * 4); e.h2d(d_idx, idx)
for _ in range(reps // 2): e.forward_device(d_idx, 1, n, l, d_out)
e.synchronize()
t0 = time.perf_counter()
for _ in range(reps): e.forward_device(d_idx, 1, n, l, d_out)
e.synchronize()
dt = (time.perf_counter() - t0) / reps
print(f"{n}x{l} batch 1: {dt * 1e6:.1f} us per forward, {1 / dt:.1f} alignments/s")
