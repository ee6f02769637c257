#!/bin/bash
# Counter passes of the ray-cast kernel for one GMUPT_TRAVERSAL value (A/B work).  Usage: tools/pmc_ab.sh <mode> [extra env assignments are inherited]
# -> gpurun_out/pmc_<mode>/pmc_counters.txt (per-kernel averages of the steady-state dispatches, tools/pmc_table.py)
export TMPDIR=/tmp
MODE=${1:?mode}
export GMUPT_TRAVERSAL=$MODE
OUT=gpurun_out/pmc_$MODE
rm -rf $OUT; mkdir -p $OUT
CMD="python3 bench.py --no-cpu-baseline --no-roofline --no-full-frame --no-config5 --prewarm 260 --steps 20 --warmup 5"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES" \
           "SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR" \
           "TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES" \
           "TCC_HIT TCC_MISS TCC_REQ TCC_EA0_RDREQ" \
           "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $OUT/pmc/pass$i -- $CMD > $OUT/pass$i.log 2>&1 || { echo "pmc pass $i ($set) failed"; tail -3 $OUT/pass$i.log; exit 1; }
done
python3 tools/pmc_table.py $OUT/pmc 265 > $OUT/pmc_counters.txt
rm -rf $OUT/pmc
cat $OUT/pmc_counters.txt
