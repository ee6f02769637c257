"""Timing experiment: the fused ray-cast launch repeated on a FIXED set of queues (snapshot of the steady state), for builds with
parts of the kernel knocked out (GMUPT_KNOCKOUT builds give wrong results; only their time is of interest)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, gmupt_pkg
g = gmupt_pkg.load(); capi = g.capi
scene = g.scenes.build_scene(g.scenes.spheres_mesh(202, 3, seed=1234))
dev = capi.Device(0); sb = capi.SceneBuffers(dev, scene)
r = capi.Renderer(dev, 1920, 1080, tile=(0, 0)); r.bind_scene(sb)
cam = capi.Camera(1920, 1080); cam.set_pose(*scene["camera"])
os.environ.pop("GMUPT_LIB_UNUSED", None)
for _ in range(300): cam.update(0.0); r.set_camera(cam.buffer); r.run_stage(capi.STAGE_SHADE); r.run_stage(capi.STAGE_EXTEND); r.run_stage(capi.STAGE_SHADOW)  # un-knocked kernels
cam.update(0.0); r.set_camera(cam.buffer); r.run_stage(capi.STAGE_SHADE)
for _ in range(3): r.run_stage(capi.STAGE_RAYCASTS)
r.synchronize(); t0 = time.perf_counter()
N = 30
for _ in range(N): r.run_stage(capi.STAGE_RAYCASTS)
r.synchronize(); dt = (time.perf_counter() - t0) / N
print(os.environ.get("GMUPT_LIB", "default").split("/")[-1], "cast ms (host clock, incl. the 16-byte counter memset)", round(dt * 1e3, 4))
