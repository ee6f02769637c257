#!/bin/bash
# SQ/TCP/TCC counter passes for the ray-cast kernels (no TA_* counters: that pass hung on this pool).  Usage: tools/pmc_run2.sh <outdir> [env...]
export TMPDIR=/tmp
OUT=${1:-gpurun_out/pmc}
mkdir -p $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES" \
           "SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR" \
           "TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES" \
           "TCC_HIT TCC_MISS TCC_REQ TCC_EA0_RDREQ" \
           "GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  echo "pass $i: $set"
  timeout -k 10 120 rocprofv3 --pmc $set --output-format csv -d $OUT/pass$i -- python3 bench.py --prewarm 260 --steps 10 --warmup 5 --no-cpu-baseline --no-roofline > $OUT/pass$i.log 2>&1 || echo "pass $i failed"
done
