#!/bin/bash
# Builds (if missing) and runs the hidden-loop microbenchmarks (tools/ffn3_bench.hip) and keeps the log
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for m in base movonly mfmaonly; do [ -x tools/ffn3_bench_$m ] || tools/build_ffn3.sh $m; done
( for m in base movonly mfmaonly; do echo "== mode $m"; timeout 120 tools/ffn3_bench_$m 2>&1 | grep -E "asm|lean" ; done ) > gpurun_out/ffn3.log 2>&1
cat gpurun_out/ffn3.log
