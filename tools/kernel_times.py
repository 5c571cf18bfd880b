"""Per-kernel HIP-event times of one forward at the headline shape (profile = 1: every launch bracketed).
    python tools/kernel_times.py [--batch 16] [--seqs 60] [--sites 500] [--steps 3]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from phyloformer_amd.engine import Engine
from phyloformer_amd.weights import load_weights
from phyloformer_amd.msa_sim import simulate_batch
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16); ap.add_argument("--seqs", type=int, default=60)
ap.add_argument("--sites", type=int, default=500); ap.add_argument("--steps", type=int, default=3)
a = ap.parse_args()
repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
eng = Engine(load_weights(os.path.join(repo, "models/pf.ckpt")))
idx = np.ascontiguousarray(np.resize(simulate_batch(min(a.batch, 8), a.seqs, a.sites, seed=3), (a.batch, a.seqs, a.sites)))
eng.forward(idx)
eng.set_option("profile", 1)
eng.profile_reset()
for _ in range(a.steps):
    eng.forward(idx)
tot = 0.0
for k in ("embed", "rowfin", "colstats", "colfin", "main"):
    n, ms = eng.profile_get(k)
    tot += ms / a.steps
    print(f"{k:9s} {n // a.steps:3d} launches/step  {ms / a.steps:8.3f} ms/step  {ms / max(n, 1):7.3f} ms/launch")
print(f"sum {tot:.3f} ms/step (batch {a.batch})")
