"""Option two_streams: a batch as two free-running half-batches on two streams.  Forward time with
device-resident indices, bit-identity."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from phyloformer_amd.engine import Engine
from phyloformer_amd.weights import load_weights
from phyloformer_amd.msa_sim import simulate_batch
w = load_weights("models/pf.ckpt")
for B, n, l in [(16, 60, 500), (4, 60, 2000), (64, 20, 200)]:
    idx = np.ascontiguousarray(np.resize(simulate_batch(4, n, l, seed=3), (B, n, l)))
    P = n * (n - 1) // 2
    ref = None
    for ts in (0, 1, 0, 1):
        e = Engine(w, 0)
        e.set_option("two_streams", ts)
        out = e.forward(idx)
        if ref is None: ref = out
        same = np.array_equal(ref.view(np.uint32), out.view(np.uint32))
        d_idx = e.malloc(idx.nbytes); d_out = e.malloc(B * P * 4); e.h2d(d_idx, idx)
        for _ in range(3): e.forward_device(d_idx, B, n, l, d_out)
        e.synchronize()
        reps = 10
        t0 = time.perf_counter()
        for _ in range(reps): e.forward_device(d_idx, B, n, l, d_out)
        e.synchronize()
        dt = (time.perf_counter() - t0) / reps
        t0 = time.perf_counter()
        for _ in range(reps): e.forward(idx)
        dth = (time.perf_counter() - t0) / reps
        print(f"{n}x{l} batch {B:2d} two_streams {ts}: {dt * 1e3:8.3f} ms  {B / dt:7.1f} aln/s  (host buffers, one call at a time: {B / dth:7.1f})  bits {'same' if same else 'DIFFER'}", flush=True)
        e.close()
