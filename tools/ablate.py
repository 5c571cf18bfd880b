#!/usr/bin/env python3
"""Timing of k_main with phases switched off (perf experiments; results are invalid numerically)."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from phyloformer_amd.engine import Engine
from phyloformer_amd.weights import load_weights
from phyloformer_amd.msa_sim import simulate_batch
w = load_weights(os.path.join(REPO, "models/pf.ckpt"))
e = Engine(w, 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
idx = simulate_batch(B, 60, 500, seed=3)
d_idx = e.malloc(idx.nbytes); d_out = e.malloc(B * 1770 * 4); e.h2d(d_idx, idx)
masks = [int(m) for m in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 1, 2, 3, 4, 8, 16, 28, 31]
for mask in masks:
    e.set_option("ablate", mask)
    e.forward_device(d_idx, B, 60, 500, d_out); e.synchronize()
    e.set_option("profile", 1); e.profile_reset()
    for _ in range(3):
        e.forward_device(d_idx, B, 60, 500, d_out)
    e.synchronize()
    res = {k: e.profile_get(k) for k in ("embed", "colstats", "main")}
    e.set_option("profile", 0)
    print(f"ablate={mask:2d}  main {res['main'][1] / res['main'][0]:.3f} ms/launch  colstats {res['colstats'][1] / res['colstats'][0]:.3f}  embed {res['embed'][1] / res['embed'][0]:.3f}", flush=True)
e.set_option("ablate", 0)
