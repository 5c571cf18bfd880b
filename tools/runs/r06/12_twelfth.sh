#!/bin/bash
# final evidence (the build with opaque split inputs): suite, profile round, studies, MHA / CLI refresh
cd "$(dirname "$0")/../../.."
O=gpurun_out/r06u; mkdir -p $O
timeout 1700 python -m pytest tests -m gpu -q --maxfail=20 > $O/pytest.txt 2>&1; grep -E "passed|failed" $O/pytest.txt
cp gpurun_out/parity_errors.json $O/ 2>/dev/null
bash tools/profile_round.sh r06u > $O/profile_round.log 2>&1
python tests/dev/guard_study.py gen $O/guard_forced.npz > $O/guard_gen.txt 2>&1
PF_STUDY_ROUTED=1 python tests/dev/guard_study.py gen $O/guard_routed.npz >> $O/guard_gen.txt 2>&1
python tools/phase_prof.py > $O/phases.txt 2>&1
( python tools/mha_bench.py; python tools/mha_bench.py --rows 500 --cols 1770 ) > $O/mha_bench.txt 2>&1
( for t in "" "--trees"; do python tools/cli_bench.py --n 4096 --seqs 20 --sites 200 $t; done; for t in "" "--trees"; do python tools/cli_bench.py --n 512 --seqs 60 --sites 500 $t; done ) > $O/cli_bench.txt 2>&1
timeout 1500 python tests/dev/soak_seeds.py 1 2 3 4 5 6 > $O/soak_seeds.txt 2>&1; tail -6 $O/soak_seeds.txt
cat $O/mha_bench.txt; tail -c 300 gpurun_out/bench_r06u.json
