#!/usr/bin/env python3
"""Does the granularity of the launch sequence matter?  One batch of 16 alignments of 60 x 500, device-resident,
processed in chunks of 16 / 8 / 4 / 2 alignments (option ws_limit_mb makes pf_forward_device cut the batch; every
chunk runs its kernels breadth-first, as two half-chunks on two streams): alignments/s and k_main ms per alignment.
    python tools/chunk_sweep.py [seconds per point]"""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from phyloformer_amd.engine import Engine
from phyloformer_amd.msa_sim import simulate_batch
from phyloformer_amd.weights import load_weights


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
    B, N, L = 16, 60, 500
    P = N * (N - 1) // 2
    w = load_weights(os.path.join(REPO, "models", "pf.ckpt"))
    idx = simulate_batch(8, N, L, seed=3)
    idx = np.ascontiguousarray(idx[np.arange(B) % 8])
    with Engine(w, 0) as e:
        d_idx, d_out = e.malloc(idx.nbytes), e.malloc(B * P * 4)
        e.h2d(d_idx, idx)
        ref = None
        for rep in range(2):
            for mb, label in ((24576, "16"), (2300, "8"), (1200, "4"), (620, "2"), (24576, "16")):
                e.set_option("ws_limit_mb", mb)
                for _ in range(2):
                    e.forward_device(d_idx, B, N, L, d_out)
                e.synchronize()
                n, t0 = 0, time.perf_counter()
                while time.perf_counter() - t0 < seconds:
                    for _ in range(4):
                        e.forward_device(d_idx, B, N, L, d_out)
                    e.synchronize()
                    n += 4
                dt = time.perf_counter() - t0
                out = np.empty((B, P), np.float32)
                e.d2h(out, d_out)
                if ref is None:
                    ref = out.copy()
                print(f"chunk <= {label:>2s} alignments (ws_limit_mb {mb:5d}): {B * n / dt:8.2f} alignments/s   same bits: {np.array_equal(out, ref)}", flush=True)
        e.free(d_idx); e.free(d_out)


main()
