#!/bin/bash
# One-stop profile of the current build: bench JSON, rocprofv3 kernel stats, HBM traffic (two PMC passes).  Usage: tools/profile_round.sh <name>
export TMPDIR=/tmp
NAME=${1:-r01_current}
OUT=gpurun_out/$NAME
mkdir -p $OUT
python3 bench.py > $OUT/bench.log 2>&1; tail -1 $OUT/bench.log > $OUT/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu-baseline --no-roofline > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py --prewarm 260 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py --prewarm 260 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $OUT/write.log 2>&1
grep -v '^W\|^I\|^E' $OUT/trace.log | tail -1 > $OUT/bench_under_rocprof.json
cp $OUT/trace/*/*kernel_stats.csv $OUT/kernel_stats.csv
python3 tools/pmc_summary.py $OUT/fetch $OUT/write $OUT/pmc_traffic.json > /dev/null
ls $OUT
