"""k_colstats2 (MFMA) vs k_colstats (VALU, default); option colstats_mfma selects the former per-launch times at batch 16 and 1."""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from phyloformer_amd.engine import Engine
from phyloformer_amd.weights import load_weights
from phyloformer_amd.msa_sim import simulate_batch
w = load_weights("models/pf.ckpt")
for B in (16, 1):
    idx = np.ascontiguousarray(np.resize(simulate_batch(min(B, 8), 60, 500, seed=3), (B, 60, 500)))
    for mfma in (0, 1):
        e = Engine(w, 0)
        e.set_option("colstats_mfma", mfma)
        e.forward(idx)
        e.set_option("profile", 1); e.profile_reset()
        for _ in range(3): e.forward(idx)
        n, ms = e.profile_get("colstats"); n2, ms2 = e.profile_get("colfin")
        print(f"batch {B:2d} {'MFMA kernel' if mfma else 'VALU kernel'}: colstats {ms / n:7.3f} ms/launch, colfin {ms2 / n2:7.3f} ms/launch")
        e.close()
