#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc CSV output per kernel: mean counter value per dispatch."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "pass*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name", "?").split("(")[0]
            acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
lines = []
for k in sorted(acc):
    lines.append(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        lines.append(f"    {c:28s} n={len(v):3d} mean={sum(v) / len(v):16.1f}")
txt = "\n".join(lines)
print(txt)
open(os.path.join(out, "pmc_summary.txt"), "w").write(txt + "\n")
