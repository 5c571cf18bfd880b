#!/usr/bin/env python3
"""Throughput at the other BASELINE.json configurations (device-resident inputs, 1 GPU)."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from phyloformer_amd.engine import Engine
from phyloformer_amd.weights import load_weights
from phyloformer_amd.msa_sim import simulate_batch

def run(ckpt, n, l, B, gaps=False, reps=3):
    e = Engine(load_weights(os.path.join(REPO, "models", ckpt)), 0)
    base = simulate_batch(min(B, 4), n, l, seed=2, gaps=gaps)
    idx = np.ascontiguousarray(base[np.arange(B) % base.shape[0]])
    P = n * (n - 1) // 2
    d_idx = e.malloc(idx.nbytes); d_out = e.malloc(B * P * 4); e.h2d(d_idx, idx)
    e.forward_device(d_idx, B, n, l, d_out); e.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        e.forward_device(d_idx, B, n, l, d_out)
    e.synchronize()
    dt = (time.perf_counter() - t0) / reps
    tok = B * P * l
    print(f"{ckpt:14s} {n:3d} x {l:4d}  batch {B:5d}: {dt * 1e3:9.2f} ms/step  {B / dt:10.1f} aln/s  "
          f"{tok / dt / 1e9:6.2f} Gtoken/s  ({602240 * tok / dt / 1e12:6.1f} TFLOP/s algorithmic)", flush=True)
    e.free(d_idx); e.free(d_out); e.close()

run("pf.ckpt", 20, 200, 1)
run("pf.ckpt", 20, 200, 64)
run("pf.ckpt", 20, 200, 1024)
run("pf.ckpt", 60, 500, 1)
run("pf.ckpt", 60, 500, 16)
run("pf.ckpt", 60, 2000, 4)
run("pf_indel.ckpt", 200, 500, 2, gaps=True)
