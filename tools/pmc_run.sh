#!/bin/bash
# Collects rocprofv3 PMC passes for the traversal kernels (each pass re-runs a short bench).  Usage: tools/pmc_run.sh <outdir>
export TMPDIR=/tmp
OUT=${1:-gpurun_out/pmc}
mkdir -p $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES" \
           "SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR" \
           "TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES" \
           "TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TA_FLAT_READ_WAVEFRONTS" \
           "TCC_HIT TCC_MISS TCC_REQ TCC_EA0_RDREQ" \
           "GRBM_GUI_ACTIVE GRBM_TA_BUSY"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/pass$i -- python3 bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-roofline > $OUT/pass$i.log 2>&1 || echo "pass $i failed"
done
ls $OUT
