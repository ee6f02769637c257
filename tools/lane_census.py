"""Where does the fused ray-cast launch (the shipped kernel: k_cast_w, or GMUPT_TRAVERSAL=cast0) lose time?  Lane census, wave lifetimes,
share of the drain and the ray-length distribution from the counting instantiation (collect_stats: several times slower, so read the
lifetimes as proportions) on the bench scene."""
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
N = 5
for _ in range(N): cam.update(0.0); r.set_camera(cam.buffer); r.iterate()
st = r.stats(); h = np.array(list(st.lane_census), dtype=np.float64)
print("lane census per loop iteration (idle, walking, stalled-with-full-FIFO, walk-done-leaves-pending):", np.round(h / max(h.sum(), 1), 3).tolist())
print("wave loop iterations per launch and wave:", h.sum() / 64 / max(st.cast_waves, 1))
print("inner SIMD eff", st.ext_inner / max(64 * st.ext_wave_inner, 1), "tri SIMD eff", st.ext_tris / max(64 * st.ext_wave_tris, 1))
print("waves per launch", st.cast_waves / N, "mean wave lifetime us", st.cast_wave_ticks / max(st.cast_waves, 1) / 100.0, "max us", st.cast_wave_ticks_max / 100.0)
print("drain (both queues empty): fraction of the wave lifetime", st.cast_drain_ticks / max(st.cast_wave_ticks, 1), "loop iterations per wave", st.cast_drain_iters / max(st.cast_waves, 1),
      "of", h.sum() / 64 / max(st.cast_waves, 1), "busy lanes per drain iteration", st.cast_drain_busy_lanes / max(st.cast_drain_iters, 1))
print("wave lifetimes, 50-us buckets (per launch):", (np.array(list(st.cast_wave_end_hist)) / N).round(0).astype(int).tolist())
rh = np.array(list(st.ray_inner_hist), dtype=np.float64)
print("extension rays by inner nodes visited, 16 per bucket (fraction):", np.round(rh / max(rh.sum(), 1), 4).tolist())
