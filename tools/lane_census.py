import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, gmupt_pkg
g = gmupt_pkg.load(); capi = g.capi
scene = g.scenes.build_scene(g.scenes.spheres_mesh(202, 3, seed=1234))
dev = capi.Device(0); sb = capi.SceneBuffers(dev, scene)
r = capi.Renderer(dev, 1920, 1080, tile=(0, 0), collect_stats=True); r.bind_scene(sb)
cam = capi.Camera(1920, 1080); cam.set_pose(*scene["camera"])
for _ in range(300): cam.update(0.0); r.set_camera(cam.buffer); r.iterate()
r.reset_stats()
for _ in range(5): cam.update(0.0); r.set_camera(cam.buffer); r.iterate()
st = r.stats(); h = np.array(list(st.ext_depth_hist), dtype=np.float64)[28:32]
print("lane census per loop iteration (idle, walking, stalled-with-full-FIFO, walk-done-leaves-pending):", np.round(h / h.sum(), 3).tolist())
print("inner SIMD eff", st.ext_inner / (64 * st.ext_wave_inner), "tri SIMD eff", st.ext_tris / (64 * st.ext_wave_tris), "wave inner iters", st.ext_wave_inner, "wave tri iters", st.ext_wave_tris)
