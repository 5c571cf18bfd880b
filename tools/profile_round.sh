#!/bin/bash
# Collect the per-round evidence on the GPU box (through gpurun): kernel statistics, PMC passes, bench line.
#   usage: bash tools/profile_round.sh r01e      -> gpurun_out/{stats,pmc,bench}_<tag>*
# Copy the summaries you want judged into profiles/ afterwards (see profiles/README.md).
set -u
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/stats_$TAG
# serial launches (the batch on one stream): the per-kernel averages agree with roofline.avg_launch_ms
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stats_$TAG -- python3 $R/bench.py --one-stream --steps 5 --warmup 2 --no-cpu-baseline --no-configs --no-parity > $R/gpurun_out/stats_$TAG.log 2>&1
# the default command: two-stream passes (half-batch launches that share the chip) + the one-stream roofline region
mkdir -p $R/gpurun_out/stats2_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stats2_$TAG -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs --no-parity > $R/gpurun_out/stats2_$TAG.log 2>&1
cd $R
bash tools/pmc.sh pmc_$TAG > gpurun_out/pmc_$TAG.log 2>&1
python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
tail -c 1500 gpurun_out/bench_$TAG.json
