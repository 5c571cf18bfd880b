#!/bin/bash
# Joules per 32-token tile of the FFN hidden loop, one instruction class at a time.
# Every variant runs back to back for SECONDS (default 3) between two reads of the socket energy counter.
#   tools/energy_table.sh [seconds]   ->  gpurun_out/energy_table.txt
cd "$(dirname "$0")/.."
S=${1:-3}
mkdir -p gpurun_out
MODES="mfmaonly_nolds mfmaonly movonly fmaonly noexp nosplit nodot poly3 base"
for m in $MODES; do [ -x tools/ffn3_bench_$m ] || tools/build_ffn3.sh ${m/_nolds/+nolds}; done
(
  for m in $MODES; do echo "== $m"; timeout 120 tools/ffn3_bench_$m energy $S 2>&1 | grep -E "^energy"; done
  echo "== base, plain C++ loop with two waves per SIMD as well"; PF_PLAIN=1 timeout 120 tools/ffn3_bench_base energy $S 2>&1 | grep -E "^energy"
  echo "== mfmaonly, all-zero weights (data-dependent part of the matrix pipe's power)"; PF_ZERO=1 timeout 120 tools/ffn3_bench_mfmaonly energy $S 2>&1 | grep -E "^energy"
  echo "== base, all-zero weights"; PF_ZERO=1 timeout 120 tools/ffn3_bench_base energy $S 2>&1 | grep -E "^energy"
) > gpurun_out/energy_table.txt 2>&1
cat gpurun_out/energy_table.txt
