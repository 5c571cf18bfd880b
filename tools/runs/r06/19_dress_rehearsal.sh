#!/bin/bash
# what the driver runs at round end, on the committed state: the GPU suite with -x, smoke(), the default bench line
cd "$(dirname "$0")/../../.."
O=gpurun_out/r06E; mkdir -p $O
timeout 1700 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
python bench.py > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
d = json.loads(open('gpurun_out/r06E/bench.json').read().strip().splitlines()[-1])
print(d['metric'], d['value'], d['unit'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline'], d.get('max_abs_err'))
PY
