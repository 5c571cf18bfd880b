#!/bin/bash
# FETCH_SIZE of k_main for two library builds (A/B inside one gpurun call): tools/fetch_ab.sh libA.so libB.so
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for lib in "$@"; do
  rm -rf $R/gpurun_out/fetch_ab; mkdir -p $R/gpurun_out/fetch_ab
  PHYLOFORMER_AMD_LIB=$R/phyloformer_amd/$lib rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/fetch_ab -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-power > $R/gpurun_out/fetch_ab.log 2>&1
  python3 - "$lib" $R/gpurun_out/fetch_ab <<'PY'
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[2], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] == "FETCH_SIZE": acc[row["Kernel_Name"].split("(")[0]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    if "k_main" in k or "k_colstats" in k: print(f"{sys.argv[1]:24s} {k:34s} FETCH_SIZE mean {sum(acc[k]) / len(acc[k]) / 1048576:7.3f} GiB (n={len(acc[k])})")
PY
done
