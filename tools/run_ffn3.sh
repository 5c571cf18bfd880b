# builds nothing: run the prebuilt tools/ffn3_bench_<mode> binaries (see tools/ffn3_bench.hip) and keep the log
mkdir -p gpurun_out
( for m in base movonly mfmaonly; do echo "== mode $m"; timeout 120 tools/ffn3_bench_$m 2>&1 | grep -E "asm|lean" ; done ) > gpurun_out/ffn3.log 2>&1
cat gpurun_out/ffn3.log
