#!/bin/bash
# final evidence of round 6 on the final kernels: kernel statistics, PMC passes (k_main traffic tied to the kernel hash),
# the bench line, k_colstats' counters for both prefetch variants, the parity table of the full suite
cd "$(dirname "$0")/../../.."
O=gpurun_out/r06k; mkdir -p $O
bash tools/profile_round.sh r06k > $O/profile_round.log 2>&1
bash tools/pmc_colstats.sh r06k/pmc_colstats > $O/pmc_colstats.log 2>&1
timeout 1700 python -m pytest tests -m gpu -q --maxfail=20 > $O/pytest.txt 2>&1
grep -E "passed|failed" $O/pytest.txt
cp gpurun_out/parity_errors.json $O/ 2>/dev/null
python tools/lone_profile.py 60 500 50 > $O/lone.txt 2>&1
tail -c 400 gpurun_out/bench_r06k.json
