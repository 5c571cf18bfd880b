#!/bin/bash
# last GPU call of round 6 on the committed state: the full GPU suite, smoke(), the bench line, and the soak with
# twenty-four seeds the round had not used (19-42)
cd "$(dirname "$0")/../../.."
O=gpurun_out/r06x; mkdir -p $O
timeout 1700 python -m pytest tests -m gpu -q --maxfail=20 > $O/pytest.txt 2>&1; grep -E "passed|failed" $O/pytest.txt
cp gpurun_out/parity_errors.json $O/ 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
python bench.py > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.json
timeout 3000 python tests/dev/soak_seeds.py $(seq 19 42) > $O/soak_seeds.txt 2>&1; tail -25 $O/soak_seeds.txt
