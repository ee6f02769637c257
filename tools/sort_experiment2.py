"""How much would ray reordering buy with the fused ray cast?  Sorts the SHADOW queue (its order is free: shadowRayCast has no RNG) and the
extension queue on the host by several keys and times the fused launch (STAGE_RAYCASTS) on the frozen queues of a steady-state iteration."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
import numpy as np, gmupt_pkg, oracle_lib as O
g = gmupt_pkg.load(); capi = g.capi
big = len(sys.argv) > 1 and sys.argv[1] == "config5"
if big:
    scene = g.scenes.build_scene(g.scenes.spheres_mesh(1953, 4, seed=1234)); W, H, P, md, warm = 3840, 2160, 1 << 23, 16, 50
else:
    scene = g.scenes.build_scene(g.scenes.spheres_mesh(202, 3, seed=1234)); W, H, P, md, warm = 1920, 1080, 1 << 21, 0, 330
dev = capi.Device(0); sb = capi.SceneBuffers(dev, scene)
r = capi.Renderer(dev, W, H, pool_paths=P, tile=(0, 0), max_depth=md); r.bind_scene(sb)
cam = capi.Camera(W, H); cam.set_pose(*scene["camera"]); cam.buffer.lightCount = scene["light_count"]
for _ in range(warm): cam.update(0.0); r.set_camera(cam.buffer); r.iterate()
cam.update(0.0); r.set_camera(cam.buffer)
r.run_stage(capi.STAGE_SHADE); r.synchronize()
q = r.read_queues(); st = r.read_path_state(); qc = r.counters()
n_ext, n_sh = int(qc[7]), int(qc[6])
ext = q[3][:n_ext].copy(); sh = q[4][:n_sh].copy()
so = O.state_field(st, P, "shadowrayOrigin").view(np.float32); sd = O.state_field(st, P, "shadowrayDirection").view(np.float32)
li = O.state_field(st, P, "lightIndex")[:, 0]
ro = O.state_field(st, P, "rayOrigin").view(np.float32); rd = O.state_field(st, P, "rayDirection").view(np.float32)

def time_cast(ext_q, sh_q, reps=7):
    q2 = q.copy(); q2[3][:n_ext] = ext_q; q2[4][:n_sh] = sh_q; r.write_queues(q2); r.write_counters(qc)
    ts = []
    for _ in range(reps):
        r.write_counters(qc); r.synchronize(); t0 = time.perf_counter(); r.run_stage(capi.STAGE_RAYCASTS); r.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return "min %.3f med %.3f ms" % (min(ts), float(np.median(ts)))

def morton(cells, bits):
    out = np.zeros(len(cells), np.uint64)
    for b in range(bits):
        for a in range(3):
            out |= ((cells[:, a].astype(np.uint64) >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b + a)
    return out

def cellkey(pts, bits):
    ok = np.isfinite(pts).all(1); lo = pts[ok].min(0); hi = pts[ok].max(0)
    c = np.clip(np.nan_to_num((pts - lo) / (hi - lo + 1e-6)) * (1 << bits), 0, (1 << bits) - 1).astype(np.int64)
    return morton(c, bits)

print("rays: ext %d shadow %d" % (n_ext, n_sh))
print("baseline                      ", time_cast(ext, sh))
valid = ext != 0xFFFFFFFF
print("shadow by light               ", time_cast(ext, sh[np.argsort(li[sh], kind="stable")]))
for bits in (3, 5, 7):
    key = (li[sh].astype(np.uint64) << np.uint64(3 * bits)) | cellkey(so[sh], bits)
    print("shadow by light+morton%d      " % bits, time_cast(ext, sh[np.argsort(key, kind="stable")]))
key = cellkey(so[sh], 5)
print("shadow by morton5 only        ", time_cast(ext, sh[np.argsort(key, kind="stable")]))
e = ext[valid]
for bits in (3, 5):
    key = cellkey(ro[e], bits)
    es = ext.copy(); es[valid] = e[np.argsort(key, kind="stable")]
    print("ext by origin morton%d         " % bits, time_cast(es, sh))
octant = ((rd[e][:, 0] > 0).astype(np.uint64) | ((rd[e][:, 1] > 0).astype(np.uint64) << np.uint64(1)) | ((rd[e][:, 2] > 0).astype(np.uint64) << np.uint64(2)))
key = (cellkey(ro[e], 4) << np.uint64(3)) | octant
es = ext.copy(); es[valid] = e[np.argsort(key, kind="stable")]
ks = (li[sh].astype(np.uint64) << np.uint64(15)) | cellkey(so[sh], 5)
print("ext morton4+octant, shadow l+m5", time_cast(es, sh[np.argsort(ks, kind="stable")]))
print("random shuffle of both        ", time_cast(np.random.default_rng(0).permutation(ext), np.random.default_rng(1).permutation(sh)))
