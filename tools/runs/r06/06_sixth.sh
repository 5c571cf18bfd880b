#!/bin/bash
# Round 6 profile round: kernel statistics, PMC passes, bench line; CLI files -> files with and without --trees; the
# softmax operator; the adversarial study as the product routes it; the soak with twelve more seeds.
cd "$(dirname "$0")/../../.."
O=gpurun_out/r06g; mkdir -p $O
bash tools/profile_round.sh r06g > $O/profile_round.log 2>&1
( cd /tmp && TMPDIR=/tmp rocprofv3 --list-avail > $GRAFT_REPO_ROOT/$O/avail.txt 2>&1 )
( for t in "" "--trees"; do python tools/cli_bench.py --n 4096 --seqs 20 --sites 200 $t; done
  for t in "" "--trees"; do python tools/cli_bench.py --n 512 --seqs 60 --sites 500 $t; done
  python tools/cli_bench.py --n 4096 --seqs 20 --sites 200 --trees --extra=--python-io ) > $O/cli_bench.txt 2>&1
python -m pytest tests/test_gpu_mha.py -m gpu -q -s > $O/mha_tests.txt 2>&1
( python tools/mha_bench.py; python tools/mha_bench.py --rows 500 --cols 1770 ) > $O/mha_bench.txt 2>&1
PF_STUDY_ROUTED=1 python tests/dev/guard_study.py gen $O/guard_routed.npz > $O/guard_gen.txt 2>&1
python tests/dev/guard_study.py gen $O/guard_forced.npz >> $O/guard_gen.txt 2>&1
timeout 2400 python tests/dev/soak_seeds.py 7 8 9 10 11 12 13 14 15 16 17 18 > $O/soak_seeds.txt 2>&1
cat $O/cli_bench.txt $O/mha_bench.txt; grep "max abs err" $O/mha_tests.txt; tail -3 $O/mha_tests.txt; tail -13 $O/soak_seeds.txt
