#!/bin/bash
# Builds tools/ffn3_bench_<mode> for every mode of tools/gen_hidden_asm.py (hipcc cross-compiles without a GPU).
#   tools/build_ffn3.sh [mode ...]        default: all modes of the energy table
set -e
cd "$(dirname "$0")/.."
MODES=${@:-"base nodot nosplit poly3 noexp movonly fmaonly mfmaonly mfmaonly+nolds"}
mkdir -p build/ffn3
for m in $MODES; do
  name=${m//+/_}
  python3 tools/gen_hidden_asm.py 6 "$m" > build/ffn3/hid_$name.inc
  sh=0xbf800000u; [ "$m" = nodot ] && sh=0xffff0000u
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -DPF_SH_CONST=$sh \
        -DPF_HID_INC="\"$PWD/build/ffn3/hid_$name.inc\"" tools/ffn3_bench.hip -o tools/ffn3_bench_$name -ldl &
done
wait
ls -la tools/ffn3_bench_*
