#!/usr/bin/env python3
"""In-kernel phase timeline of k_main2 (s_memtime ticks summed over all waves)."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from phyloformer_amd.engine import Engine
from phyloformer_amd.weights import load_weights
from phyloformer_amd.msa_sim import simulate_batch
e = Engine(load_weights(os.path.join(REPO, "models/pf.ckpt")), 0)
e.set_option("main2", 1)
B = 16
idx = np.ascontiguousarray(np.resize(simulate_batch(8, 60, 500, seed=3), (B, 60, 500)))
e.forward(idx)
e.set_option("phase_prof", 1)
e.forward(idx)
out = np.empty(8, np.float32)
e._lib.pf_debug_read(e._h, b"phase_prof", out.ctypes.data, 8)
names = ["unpack loads, ctx / fragment reads", "apply + LN + split + next loads", "hidden loop (asm, both tiles)", "store + next-row statistics / head"]
pairs = B * 1770 * 16 * 6 / 2
tot = out[:4].sum()
for k, nm in enumerate(names):
    print(f"{nm:38s} {out[k]:10.1f} Mcycles  {100 * out[k] / tot:5.1f}%   {out[k] * 1e6 / pairs / 2:8.0f} cycles per tile (one wave per SIMD)")
print(f"total {tot * 1e6 / pairs / 2:.0f} cycles per tile")
