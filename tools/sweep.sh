#!/bin/bash
# A/B sweep of traversal variants (one process each, same box): prints stage times
for v in "$@"; do
  IFS=: read mode rpw <<< "$v"
  echo -n "mode=$mode rpw=$rpw  "
  GMUPT_TRAVERSAL=$mode GMUPT_RAYS_PER_WAVE=${rpw:-256} python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stage_ms'])"
done
