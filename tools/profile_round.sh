set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/stats_r01e
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stats_r01e -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/stats_r01e.log 2>&1
cd $R
bash tools/pmc.sh pmc_r01e > gpurun_out/pmc_r01e.log 2>&1
python bench.py > gpurun_out/bench_r01e.json 2> gpurun_out/bench_r01e.err
tail -c 1500 gpurun_out/bench_r01e.json
