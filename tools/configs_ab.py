"""A/B of library builds over the BASELINE configurations inside one gpurun call: bench.py's `configs` leg per .so."""
import json, os, subprocess, sys
for name in sys.argv[1:]:
    env = dict(os.environ, PHYLOFORMER_AMD_LIB=os.path.abspath(os.path.join("phyloformer_amd", name)))
    out = subprocess.run([sys.executable, "bench.py", "--steps", "10", "--warmup", "2", "--no-cpu-baseline", "--no-power"],
                         env=env, capture_output=True, text=True).stdout
    try:
        d = json.loads(out.strip().splitlines()[-1])
        row = [f"{d['value']:8.1f}"] + [f"{v['alignments_per_s']:9.1f}" for v in d["configs"].values()]
        print(f"{name:24s} headline " + " | ".join(row), flush=True)
    except Exception:
        print(name, "failed", out[-300:])
