#!/bin/bash
cd "$(dirname "$0")/../../.."
O=gpurun_out/r06i; mkdir -p $O
bash tools/pmc_colstats.sh r06i/pmc_colstats > $O/pmc_colstats.log 2>&1
timeout 1700 python -m pytest tests -m gpu -q --maxfail=20 > $O/pytest.txt 2>&1
grep -E "passed|failed" $O/pytest.txt
cp gpurun_out/parity_errors.json $O/ 2>/dev/null
python bench.py > $O/bench.json 2> $O/bench.err
tail -c 600 $O/bench.json
cat gpurun_out/r06i/pmc_colstats/colstats_counters.txt
