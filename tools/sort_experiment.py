"""How much would ray reordering buy?  Sorts the extension queue on the host by several keys and times k_extend alone."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
import numpy as np, gmupt_pkg, oracle_lib as O
g = gmupt_pkg.load(); capi = g.capi
scene = g.scenes.build_scene(g.scenes.spheres_mesh(202, 3, seed=1234))
dev = capi.Device(0); sb = capi.SceneBuffers(dev, scene)
P = 1 << 21
r = capi.Renderer(dev, 1920, 1080, tile=(0, 0)); r.bind_scene(sb)
cam = capi.Camera(1920, 1080); cam.set_pose(*scene["camera"])
for _ in range(330): cam.update(0.0); r.set_camera(cam.buffer); r.iterate()
cam.update(0.0); r.set_camera(cam.buffer)
r.run_stage(capi.STAGE_SHADE); r.synchronize()
q = r.read_queues(); st = r.read_path_state()
ext = q[3].copy()
o = O.state_field(st, P, "rayOrigin").view(np.float32); d = O.state_field(st, P, "rayDirection").view(np.float32)

def time_extend(queue, reps=5):
    q2 = q.copy(); q2[3] = queue; r.write_queues(q2)
    ts = []
    for _ in range(reps):
        r.synchronize(); t0 = time.perf_counter(); r.run_stage(capi.STAGE_EXTEND); r.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return min(ts), np.median(ts)

def morton(cells, bits):
    out = np.zeros(len(cells), np.uint64)
    for b in range(bits):
        for a in range(3):
            out |= ((cells[:, a].astype(np.uint64) >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b + a)
    return out

print("unsorted      min/med ms", time_extend(ext))
oo = o[ext]; dd = d[ext]
lo = oo.min(0); hi = oo.max(0)
octant = ((dd[:, 0] > 0).astype(np.uint64) | ((dd[:, 1] > 0).astype(np.uint64) << np.uint64(1)) | ((dd[:, 2] > 0).astype(np.uint64) << np.uint64(2)))
for bits in (3, 5, 8):
    cells = np.clip(((oo - lo) / (hi - lo + 1e-6) * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)
    key = (octant << np.uint64(3 * bits)) | morton(cells, bits)
    print("octant+morton%d min/med ms" % bits, time_extend(ext[np.argsort(key, kind="stable")]))
    key2 = (morton(cells, bits) << np.uint64(3)) | octant
    print("morton%d+octant min/med ms" % bits, time_extend(ext[np.argsort(key2, kind="stable")]))
# direction-first: quantised direction (6 bits/axis) then origin cell
dq = np.clip(((dd * 0.5 + 0.5) * 16).astype(np.int64), 0, 15)
cells = np.clip(((oo - lo) / (hi - lo + 1e-6) * 32).astype(np.int64), 0, 31)
key3 = (morton(dq, 4) << np.uint64(15)) | morton(cells, 5)
print("dir16^3+cell32^3  min/med ms", time_extend(ext[np.argsort(key3, kind="stable")]))
print("random shuffle    min/med ms", time_extend(np.random.default_rng(0).permutation(ext)))
