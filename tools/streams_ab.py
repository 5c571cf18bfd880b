"""Two-stream / one-stream / host-buffer rate of library builds, alternating, inside one GPU call: python tools/streams_ab.py libA.so libB.so"""
import json, os, subprocess, sys
for rep in range(3):
    for name in sys.argv[1:]:
        env = dict(os.environ, PHYLOFORMER_AMD_LIB=os.path.abspath(os.path.join("phyloformer_amd", name)))
        out = subprocess.run([sys.executable, "bench.py", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-power", "--no-configs", "--no-parity"], env=env, capture_output=True, text=True).stdout
        d = json.loads(out.strip().splitlines()[-1])
        print(f"{name:20s} two-stream {d['value']:8.2f}  one-stream {d['value_one_stream']:8.2f}  host {d['value_pcie_inclusive']:8.2f}  k_main {d['roofline']['avg_launch_ms']:.3f} ms", flush=True)
