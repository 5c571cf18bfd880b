#!/bin/bash
cd "$(dirname "$0")/../../.."
O=gpurun_out/r06j; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "ring_prefetch or batch_invariance or configs" > $O/pytest_a.txt 2>&1; tail -3 $O/pytest_a.txt
( for l in libphyloformer_amd.so lib_asmtake.so libphyloformer_amd.so lib_asmtake.so; do python tools/kernel_ab.py colstats $l 2>/dev/null | head -3; done ) > $O/colstats_take_ab.txt 2>&1; cat $O/colstats_take_ab.txt
PF_AB_STEPS=10 python tools/flag_compare.py libphyloformer_amd.so lib_asmtake.so libphyloformer_amd.so lib_asmtake.so > $O/ab.txt 2>&1; cat $O/ab.txt
