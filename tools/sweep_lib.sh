#!/bin/bash
# A/B of differently configured builds of libgmupt.so: args "libfile:wavesPerCU"
for v in "$@"; do
  IFS=: read lib wpc <<< "$v"
  echo -n "$v  "
  GMUPT_LIB=$PWD/gmu-path-tracer_amd/$lib GMUPT_WAVES_PER_CU=${wpc:-16} python bench.py --steps 60 --warmup 10 --prewarm 300 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['stage_ms']['extend'], d['stage_ms']['shadow'])"
done
