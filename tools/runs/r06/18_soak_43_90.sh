#!/bin/bash
# forty-eight more soak seeds on the final build (range re-check on)
cd "$(dirname "$0")/../../.."
O=gpurun_out/r06C; mkdir -p $O
timeout 5000 python tests/dev/soak_seeds.py $(seq 43 90) > $O/soak_seeds.txt 2>&1; tail -50 $O/soak_seeds.txt
