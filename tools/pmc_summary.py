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

# HBM traffic of the dominant kernel, per token, for bench.py's roofline.traffic
import json
# k_main<MODE_MID, FLAT>: since round 3 the tiling is a template parameter; the headline shape runs <1, true>
main = next((acc[k] for k in sorted(acc) if k.startswith("void pfk::k_main<1")), None)
if main and "FETCH_SIZE" in main and "WRITE_SIZE" in main and len(sys.argv) > 2:
    tokens = float(sys.argv[2])
    fetch = sum(main["FETCH_SIZE"]) / len(main["FETCH_SIZE"])
    write = sum(main["WRITE_SIZE"]) / len(main["WRITE_SIZE"])
    # KiB units; FETCH_SIZE counts 64 B per 128-B request for 16-B/lane streaming reads on gfx950 -> x2
    hbm = (2.0 * fetch + write) * 1024.0
    rec = {"kernel": "k_main<MID>", "tokens_per_launch": tokens, "FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write,
           "hbm_bytes_per_launch": hbm, "hbm_bytes_per_token": hbm / tokens,
           "correction": "hbm = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (MI355X_MICROARCH.md, HBM section)"}
    # tie the counters to the library they were taken with (bench.py refuses them for another kernel_hash)
    try:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from phyloformer_amd import engine
        b = engine.build_info()
        rec.update({"kernel_hash": b["kernel_hash"], "source_hash": b["source_hash"], "sched_strategy": b["sched_strategy"]})
    except Exception as exc:  # noqa: BLE001
        rec["kernel_hash"] = None
        print("pmc_summary: no build info:", exc)
    json.dump(rec, open(os.path.join(out, "pmc_k_main.json"), "w"), indent=1)
    print(rec)
