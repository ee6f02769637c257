#!/bin/bash
# A/B of ray-cast kernels on one box: runs bench.py (steady state only) once per GMUPT_TRAVERSAL value given, writes gpurun_out/ab_<mode>.json
# usage: tools/ab_cast.sh cast0 wide wide@wpk0 [-- extra bench args]
set -e
mkdir -p gpurun_out
modes=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do modes+=("$1"); shift; done
[ "$1" == "--" ] && shift
for spec in "${modes[@]}"; do
  m=${spec%@*}; b=""; [ "$spec" != "$m" ] && b=${spec#*@}     # mode[@experiment-build]: libgmupt_<build>.so (gmu-path-tracer_amd/build.py EXPERIMENT_BUILDS)
  lib=""; [ -n "$b" ] && lib=$PWD/gmu-path-tracer_amd/libgmupt_$b.so
  m2=$m; m=$(echo $spec | tr '@' '_')
  GMUPT_LIB=$lib GMUPT_TRAVERSAL=$m2 python bench.py --no-cpu-baseline --no-full-frame --no-config5 "$@" > gpurun_out/ab_$m.json 2> gpurun_out/ab_$m.err || { echo "bench failed for $m"; tail -5 gpurun_out/ab_$m.err; exit 1; }
  python - "$m" <<'PY'
import json, sys
m = sys.argv[1]
j = json.loads(open("gpurun_out/ab_%s.json" % m).read().strip().splitlines()[-1])
r = j.get("roofline") or {}
print("         census %s  simd %s  redo/launch %s  general %s" % (r.get("lane_census"), r.get("simd_efficiency"), r.get("redo_rays_per_launch"), r.get("general_slab_test_share")))
print("%-8s value %.3f Mpaths/s  ms/step %.4f  raycast %.4f  logic %.4f  material %.4f  kernel %s  inner/ray %s  lds_top %s" % (
    m, j["value"], j["ms_per_step"], j["stage_ms"]["raycast"], j["stage_ms"]["logic"], j["stage_ms"]["material"], r.get("kernel"), r.get("inner_per_ray"), r.get("lds_top_share_of_node_visits")))
PY
done
