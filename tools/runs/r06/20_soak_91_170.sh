#!/bin/bash
# eighty more soak seeds on the final build; the re-check tests once more after the last host-side edit
cd "$(dirname "$0")/../../.."
O=gpurun_out/r06F; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_precise.py tests/test_gpu_sharding.py -m gpu -q -k "recheck" > $O/pytest_recheck.txt 2>&1; tail -2 $O/pytest_recheck.txt
timeout 5000 python tests/dev/soak_seeds.py $(seq 91 170) > $O/soak_seeds.txt 2>&1; grep -c " 0 violations" $O/soak_seeds.txt; grep -v " 0 violations" $O/soak_seeds.txt | cut -c1-400
