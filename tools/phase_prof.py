#!/usr/bin/env python3
"""In-kernel phase timeline of k_main (s_memtime ticks summed over all waves)."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from phyloformer_amd.engine import Engine
from phyloformer_amd.weights import load_weights
from phyloformer_amd.msa_sim import simulate_batch
w = load_weights(os.path.join(REPO, "models/pf.ckpt"))
e = Engine(w, 0)
B = 8
idx = simulate_batch(B, 60, 500, seed=3)
e.forward(idx)
e.set_option("phase_prof", 1)
e.forward(idx)
out = np.empty(8, np.float32)
n = e._lib.pf_debug_read(e._h, b"phase_prof", out.ctypes.data, 8)
names = ["wait x/q (tile start)", "apply (row+col)", "LN+split+acc init+prefetch", "hidden loop (FFN)", "store+next-row / head", "pair epilogue"]
tiles = B * 1770 * 16 * 7   # 7 k_main launches incl. FIRST
tot = out[:6].sum()
for k, nm in enumerate(names):
    print(f"{nm:30s} {out[k]:10.1f} Mcycles  {100 * out[k] / tot:5.1f}%   {out[k] * 1e6 / (B * 1770 * 16 * 6):8.0f} cycles/tile/wave (6 launches)")
print("total wave-Mcycles", tot)
