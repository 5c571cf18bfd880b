"""k_main2 (one wave per SIMD, two tiles, hand-placed hidden loop) vs k_main (option main2 = 0): results and per-launch time."""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from phyloformer_amd.engine import Engine
from phyloformer_amd.weights import load_weights
from phyloformer_amd.msa_sim import simulate_batch
w = load_weights("models/pf.ckpt")
out = {}
for B in (16, 1):
    idx = np.ascontiguousarray(np.resize(simulate_batch(min(B, 8), 60, 500, seed=3), (B, 60, 500)))
    for m2 in (1, 0, 1, 0):
        e = Engine(w, 0)
        e.set_option("main2", m2)
        out[(B, m2)] = e.forward(idx)
        e.set_option("profile", 2); e.profile_reset()
        for _ in range(3): e.forward(idx)
        n, ms = e.profile_get("main")
        print(f"batch {B:2d} {'k_main2' if m2 else 'k_main '}: {ms / n:7.3f} ms/launch", flush=True)
        e.close()
    d = np.abs(out[(B, 1)] - out[(B, 0)]).max()
    print(f"batch {B}: max |k_main2 - k_main| = {d:.3e}  bit-identical: {np.array_equal(out[(B, 1)], out[(B, 0)])}")
