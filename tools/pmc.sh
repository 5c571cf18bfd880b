#!/bin/bash
# PMC passes for the bench workload (run on the GPU box): counters only, no tracing domains.
# usage: tools/pmc.sh <outdir-under-gpurun_out> [bench args...]
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BATCH=${PMC_BATCH:-16}
# --one-stream: a launch covers the whole batch (the per-dispatch means below are per full-batch launch)
ARGS="--steps 1 --warmup 1 --batch $BATCH --no-cpu-baseline --no-configs --no-parity --no-profile --one-stream $*"
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" \
           "FETCH_SIZE GRBM_GUI_ACTIVE" \
           "WRITE_SIZE" ; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/pass$i -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/pass$i.log 2>&1
done
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT $((BATCH * 1770 * 500))
