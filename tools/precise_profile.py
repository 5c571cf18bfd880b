"""A few float64-path forwards of one shape, for `rocprofv3 --kernel-trace --stats -- python3 tools/precise_profile.py N L B`."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phyloformer_amd.engine import Engine
from phyloformer_amd.msa_sim import simulate_batch
from phyloformer_amd.weights import load_weights
n, l, b = (int(v) for v in sys.argv[1:4])
w = load_weights(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "models", "pf.ckpt"))
with Engine(w, 0) as e:
    e.set_option("precise", 1)
    idx = simulate_batch(b, n, l, seed=1)
    for _ in range(3):
        e.forward(idx)
