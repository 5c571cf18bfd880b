#!/bin/bash
# final evidence after the v_fma_mix_f32 split: A/B against the v_dot2c variant, full suite, profile round, studies
cd "$(dirname "$0")/../../.."
O=gpurun_out/r06p; mkdir -p $O
PF_AB_STEPS=10 python tools/flag_compare.py libphyloformer_amd.so lib_dot2c.so lib_bf16.so libphyloformer_amd.so lib_dot2c.so lib_bf16.so > $O/ab_fmamix_dot2c_bf16.txt 2>&1; cat $O/ab_fmamix_dot2c_bf16.txt
timeout 1700 python -m pytest tests -m gpu -q --maxfail=20 > $O/pytest.txt 2>&1; grep -E "passed|failed" $O/pytest.txt
cp gpurun_out/parity_errors.json $O/ 2>/dev/null
bash tools/profile_round.sh r06p > $O/profile_round.log 2>&1
python tests/dev/guard_study.py gen $O/guard_forced.npz > $O/guard_gen.txt 2>&1
PF_STUDY_ROUTED=1 python tests/dev/guard_study.py gen $O/guard_routed.npz >> $O/guard_gen.txt 2>&1
python tools/phase_prof.py > $O/phases.txt 2>&1
timeout 1500 python tests/dev/soak_seeds.py 1 2 3 4 5 6 > $O/soak_seeds.txt 2>&1; tail -6 $O/soak_seeds.txt
tail -c 300 gpurun_out/bench_r06p.json
