"""What the float64 path costs: alignments/s with option precise = 1 against the default kernels, per shape.
    python tools/precise_bench.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phyloformer_amd.engine import Engine
from phyloformer_amd.msa_sim import simulate_batch
from phyloformer_amd.weights import load_weights

w = load_weights(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "models", "pf.ckpt"))
with Engine(w, 0) as e:
    for (n, l, b) in [(8, 8, 64), (20, 20, 64), (40, 31, 16), (5, 100, 64), (20, 200, 16), (60, 500, 2), (60, 500, 8)]:
        idx = simulate_batch(min(b, 4), n, l, seed=1)
        idx = np.ascontiguousarray(idx[np.arange(b) % idx.shape[0]])
        P = n * (n - 1) // 2
        d_idx, d_out = e.malloc(idx.nbytes), e.malloc(b * P * 4)
        e.h2d(d_idx, idx)
        row = {}
        for mode in (1, 0):
            e.set_option("precise", mode)
            e.forward_device(d_idx, b, n, l, d_out)
            e.synchronize()
            reps = 3 if mode else 20
            t0 = time.perf_counter()
            for _ in range(reps):
                e.forward_device(d_idx, b, n, l, d_out)
            e.synchronize()
            row[mode] = (time.perf_counter() - t0) / reps
        e.free(d_idx); e.free(d_out)
        tok = b * P * l
        print(f"{n:3d} x {l:4d} batch {b:3d}: float64 {row[1] * 1e3:9.3f} ms ({b / row[1]:9.1f} aln/s, "
              f"{602240 * tok / row[1] / 1e12:6.2f} TFLOP/s fp64 algorithmic), default {row[0] * 1e3:8.3f} ms ({b / row[0]:9.1f} aln/s), "
              f"ratio {row[1] / row[0]:6.1f}", flush=True)
