#!/bin/bash
cd "$(dirname "$0")/../../.."
O=gpurun_out/r06d; mkdir -p $O
python tools/kernel_ab.py colstats libphyloformer_amd.so lib_f16a.so > $O/colstats_ab_one_stream.txt 2>&1
python tools/kernel_ab.py main libphyloformer_amd.so lib_f16a.so lib_bf16.so > $O/main_ab_one_stream.txt 2>&1
python tests/dev/precise_sweep.py $O/precise_sweep.json > $O/precise_sweep.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_precise.py tests/test_gpu_parity.py -m gpu -q -x -k "fp16_operand or errors_mirror" > $O/pytest.txt 2>&1
cat $O/colstats_ab_one_stream.txt $O/main_ab_one_stream.txt; tail -5 $O/pytest.txt; grep -c "over" $O/precise_sweep.txt
