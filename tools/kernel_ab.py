"""Per-kernel A/B of library builds inside one gpurun call: python tools/kernel_ab.py <kernel> libA.so libB.so ..."""
import os, subprocess, sys
kernel = sys.argv[1]
code = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from phyloformer_amd.engine import Engine
from phyloformer_amd.weights import load_weights
from phyloformer_amd.msa_sim import simulate_batch
w = load_weights("models/pf.ckpt")
for B in (16, 1):
    idx = np.ascontiguousarray(np.resize(simulate_batch(min(B, 8), 60, 500, seed=3), (B, 60, 500)))
    e = Engine(w, 0)
    e.set_option("two_streams", 0)          # one stream: a launch covers the whole batch and has the chip to itself
    for kv in os.environ.get("PF_AB_OPTIONS", "").split():      # e.g. PF_AB_OPTIONS="colstats_ring=0"
        k, v = kv.split("="); e.set_option(k, int(v))
    e.forward(idx)
    e.set_option("profile", 1); e.profile_reset()
    for _ in range(3): e.forward(idx)
    n, ms = e.profile_get(sys.argv[1])
    print("  batch %2d %s %.4f ms/launch" % (B, sys.argv[1], ms / n))
    e.close()
'''
for rep in range(2):
    for name in sys.argv[2:]:
        env = dict(os.environ, PHYLOFORMER_AMD_LIB=os.path.abspath(os.path.join("phyloformer_amd", name)))
        print(name, flush=True)
        sys.stdout.write(subprocess.run([sys.executable, "-c", code, kernel], env=env, capture_output=True, text=True).stdout)
        sys.stdout.flush()
