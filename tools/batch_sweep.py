"""Headline step at several batch sizes, two passes (watch the order: a box warms up within a pass): python tools/batch_sweep.py"""
import json, os, subprocess, sys
for rep in range(2):
    for b in (8, 12, 16, 20, 24, 32):
        out = subprocess.run([sys.executable, "bench.py", "--batch", str(b), "--steps", str(max(8, 320 // b)), "--warmup", "3", "--no-cpu-baseline", "--no-power", "--no-configs", "--no-parity"], capture_output=True, text=True).stdout
        d = json.loads(out.strip().splitlines()[-1])
        print(f"batch {b:3d}: two-stream {d['value']:8.2f}  one-stream {d['value_one_stream']:8.2f}  k_main {d['roofline']['avg_launch_ms']:.3f} ms ({d['roofline']['frac']})", flush=True)
