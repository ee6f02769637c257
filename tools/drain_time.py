"""How long is the drain of the fused ray cast in the PLAIN kernel?  Needs the diagnostic build -DGMUPT_DRAIN_TIMING=1 (GMUPT_LIB): every wave stamps
its start, the moment it finds both queues empty and its exit; the counting build distorts these times."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import gmupt_pkg
g = gmupt_pkg.load(); capi = g.capi
scene = g.scenes.build_scene(g.scenes.spheres_mesh(202, 3, seed=1234))
dev = capi.Device(0); sb = capi.SceneBuffers(dev, scene)
pool = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 21
r = capi.Renderer(dev, 1920, 1080, pool_paths=pool, tile=(0, 0)); r.bind_scene(sb)
cam = capi.Camera(1920, 1080); cam.set_pose(*scene["camera"]); cam.buffer.lightCount = scene["light_count"]
for _ in range(600): cam.update(0.0); r.set_camera(cam.buffer); r.iterate()
r.synchronize(); r.reset_stats(); r.enable_timing(2)
N = 100
for _ in range(N): cam.update(0.0); r.set_camera(cam.buffer); r.iterate()
st = r.stats()
waves = st.cast_waves / N
print("pool %d: launch %.4f ms (HIP events); per wave: lifetime %.1f us (max %.1f), drain %.1f us = %.1f %% of the lifetime, %.1f loop iterations in the drain; waves %d"
      % (pool, st.ms_extend / st.timed_iterations, st.cast_wave_ticks / st.cast_waves / 100.0, st.cast_wave_ticks_max / 100.0, st.cast_drain_ticks / st.cast_waves / 100.0,
         100.0 * st.cast_drain_ticks / st.cast_wave_ticks, st.cast_drain_iters / st.cast_waves, waves))
import numpy as np
h = np.array(list(st.cast_wave_end_hist), dtype=np.float64) / N
print("wave lifetimes, 40-us buckets (waves per launch):", h.round(0).astype(int).tolist())
