#!/usr/bin/env python3
"""Compact table of hipcc's -Rpass-analysis=kernel-resource-usage for the library's kernels:
    python tools/kernel_resources.py [extra hipcc flags...]
Builds to a scratch file (the in-tree .so is not touched)."""
import os, re, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phyloformer_amd import build as B

def main():
    out = os.path.join(tempfile.gettempdir(), "pf_res.o")
    unit = "pf_lib.hip"                          # the default-path kernels (k_main, k_colstats, ...)
    cmd = [B.hipcc_path(), *B.COMMON, *B.UNITS[unit], *sys.argv[1:], "-c", os.path.join(B.CSRC, unit), "-o", out,
           "-Rpass-analysis=kernel-resource-usage"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode:
        sys.stderr.write(res.stderr[-4000:]); sys.exit(1)
    cur = None; rows = {}
    for ln in res.stderr.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", ln)
        if m:
            cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip(); rows[cur] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z /\[\]]+): (\d+)", ln)
        if m and cur: rows[cur][m.group(1).strip()] = int(m.group(2))
    print(f"{'kernel':70s} VGPR AGPR spill scratch occ  LDS")
    for k, r in rows.items():
        name = re.sub(r"\(.*", "", k).replace("pfk::", "").replace("void ", "")
        print(f"{name:70s} {r.get('VGPRs',0):4d} {r.get('AGPRs',0):4d} {r.get('VGPRs Spill',0):5d} {r.get('ScratchSize [bytes/lane]',0):7d} "
              f"{r.get('Occupancy [waves/SIMD]',0):3d} {r.get('LDS Size [bytes/block]',0):6d}")
main()
