#!/bin/bash
# the range re-check (option "recheck_above"): its test, the whole GPU suite, the soak seed that found the case (27) and
# the 23 other seeds of call 13 again, the 741-case study as the product routes it
cd "$(dirname "$0")/../../.."
O=gpurun_out/r06z; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_precise.py -m gpu -q -x -s -k "range_recheck" > $O/pytest_recheck.txt 2>&1; tail -4 $O/pytest_recheck.txt
timeout 1700 python -m pytest tests -m gpu -q --maxfail=20 > $O/pytest.txt 2>&1; grep -E "passed|failed" $O/pytest.txt
cp gpurun_out/parity_errors.json $O/ 2>/dev/null
timeout 3000 python tests/dev/soak_seeds.py 27 $(seq 19 26) $(seq 28 42) > $O/soak_seeds.txt 2>&1; tail -25 $O/soak_seeds.txt
PF_STUDY_ROUTED=1 python tests/dev/guard_study.py gen $O/guard_routed.npz > $O/guard_gen.txt 2>&1; tail -2 $O/guard_gen.txt
python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json
