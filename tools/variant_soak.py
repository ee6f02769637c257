"""Soak check: several traversal rungs (GMUPT_TRAVERSAL) over many full-size iterations of the bench scene must leave bit-identical
path state, queues, counters and framebuffer.  Usage: variant_soak.py [iterations] [modes...]"""
import hashlib, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, gmupt_pkg
g = gmupt_pkg.load(); capi = g.capi
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 600
modes = sys.argv[2:] or ["cast0", "cast2", "def0"]
scene = g.scenes.build_scene(g.scenes.spheres_mesh(202, 3, seed=1234))
dev = capi.Device(0); sb = capi.SceneBuffers(dev, scene)
digests = {}
for mode in modes:
    os.environ["GMUPT_TRAVERSAL"] = mode
    r = capi.Renderer(dev, 1920, 1080, tile=(0, 0)); r.bind_scene(sb)
    cam = capi.Camera(1920, 1080); cam.set_pose(*scene["camera"])
    for _ in range(iters): cam.update(0.0); r.set_camera(cam.buffer); r.iterate()
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(r.read_path_state()).tobytes()); h.update(r.counters().tobytes()); h.update(np.ascontiguousarray(r.framebuffer()).tobytes())
    q = r.read_queues(); c = r.counters(); h.update(np.ascontiguousarray(q[3][:int(c[7])]).tobytes())
    digests[mode] = h.hexdigest(); st = r.stats()
    print(mode, iters, "iterations, completed", st.paths_completed, "sha256", digests[mode][:16], flush=True)
    r.close()
assert len(set(digests.values())) == 1, "variants differ: %r" % digests
print("all identical")
