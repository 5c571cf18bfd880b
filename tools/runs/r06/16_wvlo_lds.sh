#!/bin/bash
# half of Wv' lo (4 of 8 fragments per lane) in the LDS image (the product build) against all eight from L2 (-DPF_WVLO_LDS=0)
cd "$(dirname "$0")/../../.."
O=gpurun_out/r06A; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $O/pytest_parity.txt 2>&1; tail -3 $O/pytest_parity.txt
PF_AB_STEPS=10 python tools/flag_compare.py libphyloformer_amd.so lib_wvlo0.so libphyloformer_amd.so lib_wvlo0.so libphyloformer_amd.so lib_wvlo0.so > $O/ab.txt 2>&1; cat $O/ab.txt
python tools/kernel_ab.py main libphyloformer_amd.so lib_wvlo0.so > $O/main_ab_one_stream.txt 2>&1; cat $O/main_ab_one_stream.txt
