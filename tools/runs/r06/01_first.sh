#!/bin/bash
# Round 6, first GPU call: the fp16 probe, the hidden loop's energy in both operand formats, the GPU suite on the fp16
# build, an alternating A/B of the fp16 / bf16 / round-5 builds and the 741-case adversarial study on each.
cd "$(dirname "$0")/../../.."
O=gpurun_out/r06a; mkdir -p $O
tools/f16_probe 3 > $O/f16_probe.txt 2>&1
( for r in 1 2; do for f in 0 1; do echo "== PF_F16=$f"; PF_PLAIN=1 timeout 120 tools/ffn3_bench_fmt$f energy 4 2>&1 | grep -E "^energy"; done; done ) > $O/ffn3_energy.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 --deselect tests/test_gpu_precise.py::test_soak_every_accepted_shape_within_the_reference_error > $O/pytest.txt 2>&1
cp gpurun_out/parity_errors.json $O/ 2>/dev/null
PF_AB_STEPS=10 python tools/flag_compare.py libphyloformer_amd.so lib_bf16.so lib_r05.so libphyloformer_amd.so lib_bf16.so lib_r05.so > $O/ab.txt 2>&1
python tests/dev/guard_study.py gen $O/guard_f16.npz > $O/guard_gen.txt 2>&1
PHYLOFORMER_AMD_LIB=$PWD/phyloformer_amd/lib_r05.so python tests/dev/guard_study.py gen $O/guard_r05.npz >> $O/guard_gen.txt 2>&1
tail -n 30 $O/f16_probe.txt $O/ffn3_energy.txt $O/ab.txt $O/guard_gen.txt; tail -n 40 $O/pytest.txt
