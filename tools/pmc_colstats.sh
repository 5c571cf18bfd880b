#!/bin/bash
# What bounds k_colstats?  Counter passes over the bench workload for the ring variant and the
# register-prefetch variant (PF_COLSTATS_RING=0): instruction counts, pipe activity, in-flight memory instructions
# (SQ_INST_LEVEL_VMEM / cycles = loads in flight per CU), L2 read latency (TCP_TCC_READ_REQ_LATENCY / TCP_TCC_READ_REQ).
#   usage (GPU box): tools/pmc_colstats.sh <outdir-under-gpurun_out>
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 1 --warmup 1 --batch 16 --no-cpu-baseline --no-configs --no-parity --no-profile --one-stream"
for ring in 1 0; do
  i=0
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" \
             "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
             "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_BUSY_CU_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" \
             "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
             "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_BUSY_sum" \
             "FETCH_SIZE GRBM_GUI_ACTIVE" ; do
    i=$((i+1))
    PF_COLSTATS_RING=$ring rocprofv3 --pmc $grp --output-format csv -d $OUT/ring$ring/pass$i -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/ring$ring.pass$i.log 2>&1
  done
  python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT/ring$ring > /dev/null 2>&1
  echo "== PF_COLSTATS_RING=$ring"; grep -A40 "k_colstats<false" $OUT/ring$ring/pmc_summary.txt | sed -n 1,40p
done > $OUT/colstats_counters.txt 2>&1
cat $OUT/colstats_counters.txt
