set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/stats_r01d
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stats_r01d -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/stats_r01d.log 2>&1
cd $R
bash tools/pmc.sh pmc_r01d > gpurun_out/pmc_r01d.log 2>&1
python bench.py > gpurun_out/bench_r01d.json 2> gpurun_out/bench_r01d.err
tail -c 1500 gpurun_out/bench_r01d.json
