"""Whole-frame job profile: live slots / completed paths / time as the budgeted render proceeds (where do the iterations go?)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, gmupt_pkg
g = gmupt_pkg.load(); capi = g.capi
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
W, H = 1920, 1080
scene = g.scenes.build_scene(g.scenes.spheres_mesh(202, 3, seed=1234))
dev = capi.Device(0); sb = capi.SceneBuffers(dev, scene)
r = capi.Renderer(dev, W, H, tile=(0, 0), path_budget=W * H * spp); r.bind_scene(sb)
cam = capi.Camera(W, H); cam.set_pose(*scene["camera"])
t0 = time.perf_counter(); last_seg = 0; last_t = t0
for k in range(1, 40001):
    cam.update(0.0); r.set_camera(cam.buffer); r.iterate()
    if k % 250 == 0:
        st = r.stats(); t = time.perf_counter()
        print("it %5d  active %7d  generated %9d  completed %9d  live/iter %8.0f  ms/iter %.3f" % (k, st.active_paths, st.paths_generated, st.paths_completed, (st.segments - last_seg) / 250.0, (t - last_t) / 250 * 1e3), flush=True)
        last_seg = st.segments; last_t = t
        if st.active_paths < 100: break
