"""A/B of library builds inside one gpurun call (box-to-box variance is +-3 %): bench each .so given on the command line."""
import json, os, subprocess, sys
for name in sys.argv[1:]:
    p = os.path.join("phyloformer_amd", name)
    env = dict(os.environ, PHYLOFORMER_AMD_LIB=os.path.abspath(p))
    out = subprocess.run([sys.executable, "bench.py", "--steps", os.environ.get("PF_AB_STEPS", "8"), "--warmup", "2", "--no-cpu-baseline", "--no-power", "--no-configs", "--no-parity"], env=env,
                         capture_output=True, text=True).stdout
    try:
        d = json.loads(out.strip().splitlines()[-1])
        print(f"{name:24s} {d['value']:8.2f} aln/s  k_main {d['roofline']['avg_launch_ms']:.3f} ms", flush=True)
    except Exception:
        print(name, "failed", out[-300:])
