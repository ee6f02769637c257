"""Empirical check that a result-neutral option really is neutral: two renderers (option off / on) step in lockstep at full size and
their complete path state + framebuffer are compared bit for bit every CHECK iterations.  Usage: prune_check.py ENVVAR iterations"""
import os, sys, time, zlib
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, gmupt_pkg
g = gmupt_pkg.load(); capi = g.capi
var = sys.argv[1]; iters = int(sys.argv[2]); check = int(sys.argv[3]) if len(sys.argv) > 3 else 100
scene = g.scenes.build_scene(g.scenes.spheres_mesh(202, 3, seed=1234))
dev = capi.Device(0); sb = capi.SceneBuffers(dev, scene)
rs = []
for val in ("0", "1"):
    os.environ[var] = val
    r = capi.Renderer(dev, 1920, 1080, tile=(0, 0)); r.bind_scene(sb)
    cam = capi.Camera(1920, 1080); cam.set_pose(*scene["camera"])
    rs.append((r, cam))
rays = 0; bad = 0; t0 = time.time()
for it in range(1, iters + 1):
    for r, cam in rs:
        cam.update(0.0); r.set_camera(cam.buffer); r.iterate()
    if it % check == 0:
        a = rs[0][0].framebuffer(); b = rs[1][0].framebuffer()
        sa = rs[0][0].read_path_state(); sb_ = rs[1][0].read_path_state()
        same = np.array_equal(a.view(np.uint32), b.view(np.uint32)) and np.array_equal(sa, sb_)
        bad += (not same)
        st = rs[0][0].stats()
        print("iter %d: %s  (%.1f G segments so far, %.0f s)" % (it, "identical" if same else "DIFFERENT", st.segments / 1e9, time.time() - t0), flush=True)
        if not same:
            diff = np.flatnonzero(sa.view(np.uint32) != sb_.view(np.uint32)); print("  first differing state words:", diff[:8], "count", diff.size); break
print("RESULT %s=%s" % (var, "neutral" if bad == 0 else "NOT neutral"))
