#!/bin/bash
# One-stop profile of the current build.  Usage: tools/profile_round.sh <name> config3|config5   -> gpurun_out/<name>/
#   run.json            the plain (un-profiled) measurement line: bench.py for config3, tools/config5.py --count for config5
#   kernel_stats.csv    rocprofv3 --kernel-trace --stats of the same command (average launch durations)
#   pmc_traffic.json    two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) summarised per launch by tools/pmc_summary.py
#   pmc_counters.txt    per-kernel averages of the SQ / TCP / TCC / GRBM passes below (tools/pmc_table.py)
#   ea_read_sizes.txt   the L2's memory-side read requests by size (TCC_EA0_RDREQ_{32B,64B,128B}); roofline.json = tools/roofline_summary.py over these files
# Every profiled command is the program itself after `--` (no env / bash -c hop), every pass has its own `timeout -k`, and a pass holds at most
# as many counters of a block as the block has slots (MI355X_MICROARCH.md "rocprofv3 PMC slots"; an over-subscribed pass aborts the profiled
# process: profiles/README.md "TA counter pass").
export TMPDIR=/tmp
NAME=${1:?name}; MODE=${2:-config3}
OUT=gpurun_out/$NAME
mkdir -p $OUT
if [ "$MODE" = config5 ]; then
  timeout -k 10 500 python3 tools/config5.py --count --cache /tmp/config5_scene.npz --out $OUT/run.json > $OUT/run.log 2>&1 || { echo "plain run failed"; tail -5 $OUT/run.log; exit 1; }
  CMD="python3 tools/config5.py --cache /tmp/config5_scene.npz"; SHORT="--prewarm 40 --steps 10"; SKIP=40
else
  timeout -k 10 500 python3 bench.py > $OUT/run.log 2>&1 || { echo "plain run failed"; tail -5 $OUT/run.log; exit 1; }
  tail -1 $OUT/run.log > $OUT/run.json
  CMD="python3 bench.py --no-cpu-baseline --no-roofline --no-full-frame --no-config5"; SHORT="--prewarm 260 --steps 20 --warmup 5"; SKIP=265
fi
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1 || { echo "trace pass failed"; tail -5 $OUT/trace.log; exit 1; }
cp $OUT/trace/*/*kernel_stats.csv $OUT/kernel_stats.csv
grep -v '^W\|^I\|^E' $OUT/trace.log | tail -1 > $OUT/run_under_rocprof.json
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES" \
           "TCC_HIT TCC_MISS TCC_REQ TCC_EA0_RDREQ" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES" \
           "SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR" \
           "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --pmc $set --output-format csv -d $OUT/pmc/pass$i -- $CMD $SHORT > $OUT/pmc_pass$i.log 2>&1 || { echo "pmc pass $i ($set) failed"; tail -3 $OUT/pmc_pass$i.log; exit 1; }
done
timeout -k 10 400 rocprofv3 --pmc TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_64B TCC_EA0_RDREQ_128B TCC_EA0_RDREQ_DRAM --output-format csv -d $OUT/pmc_ea/pass1 -- $CMD $SHORT > $OUT/pmc_ea.log 2>&1 || { echo "pmc pass (read request sizes) failed"; tail -3 $OUT/pmc_ea.log; exit 1; }
python3 tools/pmc_table.py $OUT/pmc_ea $SKIP > $OUT/ea_read_sizes.txt
python3 tools/pmc_summary.py $OUT/pmc/pass1 $OUT/pmc/pass2 $OUT/pmc_traffic.json $SKIP > /dev/null
python3 tools/pmc_table.py $OUT/pmc $SKIP > $OUT/pmc_counters.txt
python3 tools/roofline_summary.py $OUT > /dev/null
rm -rf $OUT/trace $OUT/pmc $OUT/pmc_ea      # the raw per-dispatch CSVs stay on the box; the summaries above are what gets committed
ls $OUT
