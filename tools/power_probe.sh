#!/bin/bash
# Samples rocm-smi (power, clocks) while the headline bench runs: is k_main power-limited?
mkdir -p gpurun_out
python3 bench.py --steps 900 --warmup 2 --no-cpu-baseline > gpurun_out/power_bench.json 2> gpurun_out/power_bench.err &
BP=$!
for i in $(seq 1 60); do
  echo -n "t=$i "; rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power \(W\)|sclk" | sed 's/.*: //' | tr '\n' ' '; echo
  sleep 0.5
done > gpurun_out/power_samples.txt
wait $BP
cat gpurun_out/power_samples.txt
cut -c1-200 gpurun_out/power_bench.json
