"""What a rank of a site-sharded weak-scaling run computes, without the collectives: a non-sharded forward of
B x world alignments of ceil(L / world) sites has the per-rank kernel shapes (tiles, column chunks, partial
counts) of `bench.py --gpus world`.  Equal token counts, so equal times mean the sharding itself costs nothing."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from phyloformer_amd.engine import Engine
from phyloformer_amd.weights import load_weights
from phyloformer_amd.msa_sim import simulate_batch
w = load_weights("models/pf.ckpt")
for L in (500, 2000):
    for world in (1, 2, 4, 8):
        B, n = (16 if L == 500 else 4) * world, 60
        l = -(-L // world)
        idx = np.ascontiguousarray(np.resize(simulate_batch(4, n, l, seed=3), (B, n, l)))
        P = n * (n - 1) // 2
        e = Engine(w, 0)
        d_idx = e.malloc(idx.nbytes); d_out = e.malloc(B * P * 4); e.h2d(d_idx, idx)
        for _ in range(2): e.forward_device(d_idx, B, n, l, d_out)
        e.synchronize()
        reps = 6
        t0 = time.perf_counter()
        for _ in range(reps): e.forward_device(d_idx, B, n, l, d_out)
        e.synchronize()
        dt = (time.perf_counter() - t0) / reps
        tok = B * P * l
        print(f"L {L:4d} world {world}: per rank {B:3d} x (60 x {l:4d})  {dt * 1e3:7.2f} ms  {tok / dt / 1e9:5.3f} Gtoken/s  "
              f"all-reduce payload per block {B * P * 72 * 4 / 1e6:6.1f} MB", flush=True)
        e.close()
