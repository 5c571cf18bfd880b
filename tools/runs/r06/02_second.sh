#!/bin/bash
# Round 6, GPU call: packed-fp32 k_colstats and the hazard-safe split8 against the first fp16 build and the bf16 build.
cd "$(dirname "$0")/../../.."
O=gpurun_out/r06c; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 --deselect tests/test_gpu_precise.py::test_soak_every_accepted_shape_within_the_reference_error > $O/pytest.txt 2>&1
cp gpurun_out/parity_errors.json $O/ 2>/dev/null
python tools/kernel_ab.py colstats libphyloformer_amd.so lib_f16a.so > $O/colstats_ab.txt 2>&1
PF_AB_STEPS=10 python tools/flag_compare.py libphyloformer_amd.so lib_f16a.so lib_bf16.so libphyloformer_amd.so lib_f16a.so lib_bf16.so > $O/ab.txt 2>&1
tail -n 12 $O/pytest.txt; cat $O/colstats_ab.txt $O/ab.txt
