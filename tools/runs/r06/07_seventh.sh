#!/bin/bash
cd "$(dirname "$0")/../../.."
O=gpurun_out/r06h; mkdir -p $O
( for l in libphyloformer_amd.so lib_r6m8.so lib_r4m8.so lib_r2m16.so libphyloformer_amd.so lib_r6m8.so; do python tools/kernel_ab.py colstats $l | head -3; done ) > $O/colstats_variants.txt 2>&1
cat $O/colstats_variants.txt
PF_AB_STEPS=10 python tools/flag_compare.py libphyloformer_amd.so lib_r6m8.so lib_r4m8.so libphyloformer_amd.so lib_r6m8.so lib_r4m8.so > $O/ab.txt 2>&1
cat $O/ab.txt
timeout 1500 python -m pytest tests/test_gpu_precise.py -m gpu -q > $O/pytest_precise.txt 2>&1; tail -3 $O/pytest_precise.txt
timeout 2400 python tests/dev/soak_seeds.py 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16 17 18 > $O/soak_seeds.txt 2>&1; tail -19 $O/soak_seeds.txt
