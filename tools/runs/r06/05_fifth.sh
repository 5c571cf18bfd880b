#!/bin/bash
# full GPU suite (soak included) with the L < 32 routing rule, then the soak with six other seeds
cd "$(dirname "$0")/../../.."
O=gpurun_out/r06f; mkdir -p $O
timeout 1700 python -m pytest tests -m gpu -q --maxfail=20 > $O/pytest.txt 2>&1
tail -15 $O/pytest.txt
cp gpurun_out/parity_errors.json $O/ 2>/dev/null
timeout 1500 python tests/dev/soak_seeds.py 1 2 3 4 5 6 > $O/soak_seeds.txt 2>&1
cat $O/soak_seeds.txt
