#!/usr/bin/env python3
"""Markdown table of the GPU parity errors (tests/test_gpu_parity.py writes gpurun_out/parity_errors.json).
    python tools/parity_table.py profiles/r03_parity_errors.json"""
import json
import sys

d = json.load(open(sys.argv[1]))
print("| case (reference output it is compared with) | max abs error | max \\|reference\\| | GPU vs fp64 | fp32 reference vs fp64 |")
print("|---|---|---|---|---|")
for k, v in sorted(d.items()):
    ref = "—" if v.get("max_abs_ref") is None else f"{v['max_abs_ref']:.3g}"
    g64 = f"{v['gpu_vs_fp64']:.2e}" if "gpu_vs_fp64" in v else ""
    r64 = f"{v['fp32_reference_vs_fp64']:.2e}" if "fp32_reference_vs_fp64" in v else ""
    name = k
    if "detail" in v:
        name += " (" + ", ".join(f"{n} {x:.1e}" for n, x in sorted(v["detail"].items())) + "; relative to the largest entry)"
    print(f"| {name} | {v['max_abs_err']:.2e} | {ref} | {g64} | {r64} |")
