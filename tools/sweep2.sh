#!/bin/bash
# sweep of the deferred-leaf kernels: args "mode:rpw:refill:thresh"
for v in "$@"; do
  IFS=: read mode rpw refill thresh <<< "$v"
  echo -n "$v  "
  GMUPT_TRAVERSAL=$mode GMUPT_RAYS_PER_WAVE=${rpw:-512} GMUPT_REFILL=${refill:-20} GMUPT_TRI_THRESH=${thresh:-40} python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stage_ms']['extend'], d['stage_ms']['shadow'])"
done
