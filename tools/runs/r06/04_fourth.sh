#!/bin/bash
cd "$(dirname "$0")/../../.."
O=gpurun_out/r06e; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_precise.py -m gpu -q -x -k "ring_prefetch or fp16_operand or errors_mirror or batch_invariance or configs" > $O/pytest_a.txt 2>&1
tail -5 $O/pytest_a.txt
( echo "== ring"; python tools/kernel_ab.py colstats libphyloformer_amd.so; echo "== registers"; PF_AB_OPTIONS="colstats_ring=0" python tools/kernel_ab.py colstats libphyloformer_amd.so ) > $O/colstats_ring_ab.txt 2>&1
cat $O/colstats_ring_ab.txt
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-configs > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline'], d.get('max_abs_err'))"
