#!/bin/bash
# config-5 A/B on one box: tools/c5_ab.sh mode[@build] ...   (scene cached in /tmp on the box)
for spec in "$@"; do
  m=${spec%@*}; b=""; [ "$spec" != "$m" ] && b=${spec#*@}
  lib=""; [ -n "$b" ] && lib=$PWD/gmu-path-tracer_amd/libgmupt_$b.so
  tag=$(echo $spec | tr '@' '_')
  GMUPT_LIB=$lib GMUPT_TRAVERSAL=$m timeout -k 10 400 python tools/config5.py --count --cache /tmp/c5.npz --out gpurun_out/c5_$tag.json > gpurun_out/c5_$tag.log 2>&1 || { tail -5 gpurun_out/c5_$tag.log; exit 1; }
  python - $tag <<'PY'
import json, sys
j = json.loads(open("gpurun_out/c5_%s.json" % sys.argv[1]).read())
c = j["cast"]
print("%-14s ms/step %.3f  raycast %.4f  logic %.3f  material %.3f  inner/ray %.2f  lds_top %.3f" % (sys.argv[1], j["ms_per_step"], c["avg_launch_ms"], j["stage_ms"]["logic"], j["stage_ms"]["material"], c["inner_per_ext_ray"], c["lds_top_share_of_node_visits"]), flush=True)
PY
done
