import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from phyloformer_amd.engine import Engine
from phyloformer_amd.weights import load_weights
from phyloformer_amd.msa_sim import simulate_batch
e = Engine(load_weights("models/pf.ckpt"), 0)
idx = np.ascontiguousarray(np.resize(simulate_batch(8, 60, 500, seed=3), (16, 60, 500)))
e.forward(idx)
for ab, name in ((0, "full"), (128, "stage only the first tile (math only)"), (256, "no math (staging only)"), (384, "neither")):
    e.set_option("ablate", ab)
    e.set_option("profile", 1); e.profile_reset()
    for _ in range(2): e.forward(idx)
    n, ms = e.profile_get("colstats")
    print(f"{name:42s} {ms / n:7.3f} ms/launch")
    e.set_option("profile", 0)
