"""BASELINE config 5: ~10 M-triangle scene, 3840x2160, pool 2^23, max depth 16 -- HBM-bound stress run (steady-state stage times)."""
import json, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, gmupt_pkg
g = gmupt_pkg.load(); capi = g.capi
n_spheres = int(sys.argv[1]) if len(sys.argv) > 1 else 1953
t = time.time(); mesh = g.scenes.spheres_mesh(n_spheres, 4, seed=1234); print("generated %d triangles in %.1f s" % (mesh["indices"].shape[0], time.time() - t), flush=True)
t = time.time(); scene = g.scenes.build_scene(mesh); build_s = time.time() - t
print("SBVH build + flatten %.1f s: %d nodes, %d references, depth %d, SAH %.1f" % (build_s, scene["nodes"].shape[0], scene["tris"].shape[0], scene["depth"], scene["sah"]), flush=True)
dev = capi.Device(0); t = time.time(); sb = capi.SceneBuffers(dev, scene)
W, H, P = 3840, 2160, 1 << 23
r = capi.Renderer(dev, W, H, pool_paths=P, tile=(0, 0), max_depth=16, collect_stats=False); r.bind_scene(sb); print("upload + bind %.1f s" % (time.time() - t), flush=True)
cam = capi.Camera(W, H); cam.set_pose(*scene["camera"])
def step(n):
    for _ in range(n): cam.update(0.0); r.set_camera(cam.buffer); r.iterate()
step(60); r.synchronize(); r.reset_stats(); r.enable_timing(True)
t = time.perf_counter(); step(40); r.synchronize(); dt = time.perf_counter() - t
st = r.stats()
out = {"config": "config5", "triangles": int(mesh["indices"].shape[0]), "nodes": int(scene["nodes"].shape[0]), "depth": scene["depth"], "build_s": round(build_s, 1),
       "ms_per_step": round(dt / 40 * 1e3, 3), "mpaths_per_s": round(st.paths_completed / dt / 1e6, 2), "msegments_per_s": round(st.segments / dt / 1e6, 1),
       "stage_ms": {k: round(getattr(st, "ms_" + k) / st.timed_iterations, 3) for k in ("logic", "scan", "material", "extend", "shadow")},
       "stack_overflow": (st.flags & 1)}
print(json.dumps(out), flush=True)
