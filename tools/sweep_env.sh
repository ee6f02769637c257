#!/bin/bash
# A/B by environment: each argument is a space-free "VAR=val,VAR=val" list
for v in "$@"; do
  echo -n "$v  "
  env $(echo $v | tr ',' ' ') python bench.py --steps 201 --warmup 10 --prewarm 400 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stage_ms'])"
done
