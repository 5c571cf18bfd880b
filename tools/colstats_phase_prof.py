#!/usr/bin/env python3
"""In-kernel phase timeline of k_colstats2 (s_memtime ticks summed over all waves; option ablate = 64)."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from phyloformer_amd.engine import Engine
from phyloformer_amd.weights import load_weights
from phyloformer_amd.msa_sim import simulate_batch
w = load_weights(os.path.join(REPO, "models/pf.ckpt"))
e = Engine(w, 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
idx = simulate_batch(min(B, 8), 60, 500, seed=3)
idx = np.ascontiguousarray(idx[np.arange(B) % idx.shape[0]])
e.forward(idx)
e.set_option("ablate", 64)
e.set_option("phase_prof", 1)
e.forward(idx)
out = np.empty(8, np.float32)
e._lib.pf_debug_read(e._h, b"phase_prof", out.ctypes.data, 8)
names = ["wait for the staged tile", "LDS reads + row apply (6 MFMA)", "LayerNorm", "split + q/k MFMA (12) + elu", "q' store", "Z update (128 FMA)"]
iters = B * 1770 * 16 * 6          # wave iterations (32 tokens each), 6 launches
tot = out[:6].sum()
for k, nm in enumerate(names):
    print(f"{nm:34s} {out[k]:10.1f} Mcycles  {100 * out[k] / tot:5.1f}%   {out[k] * 1e6 / iters:8.0f} cycles per wave iteration")
print(f"total {tot * 1e6 / iters:.0f} cycles per wave iteration (two waves share a SIMD)")
