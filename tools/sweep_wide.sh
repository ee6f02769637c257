#!/bin/bash
# sweeps the runtime scheduling knobs of the wide ray cast on one box: tools/sweep_wide.sh <build or ""> "<refill list>" "<tri thresh list>" [extra bench args]
B=$1; RF=$2; TT=$3; shift 3
lib=""; [ -n "$B" ] && lib=$PWD/gmu-path-tracer_amd/libgmupt_$B.so
for rf in $RF; do for tt in $TT; do
  GMUPT_LIB=$lib GMUPT_TRAVERSAL=wide GMUPT_REFILL=$rf GMUPT_TRI_THRESH=$tt python bench.py --no-cpu-baseline --no-full-frame --no-config5 --no-roofline --prewarm 1005 "$@" > /tmp/sw.json 2>/tmp/sw.err || { tail -3 /tmp/sw.err; exit 1; }
  python - $rf $tt <<'PY'
import json, sys
j = json.loads(open("/tmp/sw.json").read().strip().splitlines()[-1])
print("refill %s tri_thresh %s: raycast %.4f ms  value %.3f" % (sys.argv[1], sys.argv[2], j["stage_ms"]["raycast"], j["value"]), flush=True)
PY
done; done
