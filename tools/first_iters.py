import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, gmupt_pkg
g = gmupt_pkg.load(); capi = g.capi
scene = g.scenes.build_scene(g.scenes.spheres_mesh(202, 3, seed=1234))
dev = capi.Device(0); sb = capi.SceneBuffers(dev, scene)
r = capi.Renderer(dev, 1920, 1080, tile=(0, 0)); r.bind_scene(sb)
cam = capi.Camera(1920, 1080); cam.set_pose(*scene["camera"])
r.enable_timing(True)
prev = None
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    cam.update(0.0); r.set_camera(cam.buffer); r.iterate()
    st = r.stats(); qc = r.counters()
    cur = np.array([st.ms_logic, st.ms_scan, st.ms_material, st.ms_extend, st.ms_shadow])
    d = cur - (prev if prev is not None else 0); prev = cur
    print(it, "shadow rays", int(qc[6]), "ms logic/scan/mat/ext/sh:", np.round(d, 3).tolist(), "completed", st.paths_completed)
