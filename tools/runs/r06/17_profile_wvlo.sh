#!/bin/bash
# profile round + full GPU suite on the build with half of Wv' lo in LDS (the kernels' final state: new kernel_hash,
# so pmc_k_main.json is collected again); the large-shape study as routed and with the default kernels forced
cd "$(dirname "$0")/../../.."
O=gpurun_out/r06B; mkdir -p $O
timeout 1700 python -m pytest tests -m gpu -q --maxfail=20 > $O/pytest.txt 2>&1; grep -E "passed|failed" $O/pytest.txt
cp gpurun_out/parity_errors.json $O/ 2>/dev/null
bash tools/profile_round.sh r06B > $O/profile_round.log 2>&1
python tools/phase_prof.py > $O/phases.txt 2>&1
python tests/dev/large_shape_study.py gen $O/large_routed.npz > $O/large_gen.txt 2>&1
PF_STUDY_FORCED=1 python tests/dev/large_shape_study.py gen $O/large_forced.npz >> $O/large_gen.txt 2>&1; cat $O/large_gen.txt
PF_STUDY_ROUTED=1 python tests/dev/guard_study.py gen $O/guard_routed.npz > $O/guard_gen.txt 2>&1
python tests/dev/guard_study.py gen $O/guard_forced.npz >> $O/guard_gen.txt 2>&1
timeout 1500 python tests/dev/soak_seeds.py 1 2 3 4 5 6 > $O/soak_seeds.txt 2>&1; tail -6 $O/soak_seeds.txt
tail -c 300 gpurun_out/bench_r06B.json
