"""Does an L2 per part of the tree pay?  (round-3 review, item 7)

Each of the 8 XCDs has its own 4 MB L2, and with rays in queue order all eight cache the same hot part of the 17 MB table (config 3) /
0.7 GB table (config 5).  This harness freezes the queues of a steady-state iteration, bins the rays of both queues on the host into
eight bins by a spatial key (slot order kept inside a bin, bins padded to one length with queue holes) and times the wide ray cast of the
`wxcd` experiment build, in which a workgroup serves the queue segment of ITS XCD first (HW_REG_XCC_ID) and the others when that is empty.
Three timings on one box: queue order as it is; binned queues with the XCD rule off (what the permutation alone costs: ray-state loads are no
longer coalesced across a chunk's neighbours); binned queues with the XCD rule on.  With --pmc-mode only one of them runs (for rocprofv3 --pmc).

  python3 tools/xcd_experiment.py [config5] [--key octant|slabx|dir] [--pmc-mode base|binned|xcd]
"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
os.environ["GMUPT_LIB"] = os.path.join(R, "gmu-path-tracer_amd", "libgmupt_wxcd.so")
os.environ["GMUPT_TRAVERSAL"] = "wide"
import numpy as np, gmupt_pkg, oracle_lib as O
g = gmupt_pkg.load(); capi = g.capi
big = "config5" in sys.argv
key_name = sys.argv[sys.argv.index("--key") + 1] if "--key" in sys.argv else "octant"
pmc_mode = sys.argv[sys.argv.index("--pmc-mode") + 1] if "--pmc-mode" in sys.argv else None
if big:
    scene = g.scenes.build_scene(g.scenes.spheres_mesh(1953, 4, seed=1234)); W, H, P, md, warm = 3840, 2160, 1 << 23, 16, 50
else:
    scene = g.scenes.build_scene(g.scenes.spheres_mesh(202, 3, seed=1234)); W, H, P, md, warm = 1920, 1080, 1 << 21, 0, 330
dev = capi.Device(0); sb = capi.SceneBuffers(dev, scene)


def renderer(xcd):
    os.environ["GMUPT_XCD_BINS"] = "1" if xcd else "0"
    r = capi.Renderer(dev, W, H, pool_paths=P, live_paths=P - 4096, tile=(0, 0), max_depth=md); r.bind_scene(sb)   # (a little room in the queues for the padding of the bins)
    return r


r = renderer(False)
cam = capi.Camera(W, H); cam.set_pose(*scene["camera"]); cam.buffer.lightCount = scene["light_count"]
for _ in range(warm): cam.update(0.0); r.set_camera(cam.buffer); r.iterate()
cam.update(0.0); r.set_camera(cam.buffer)
r.run_stage(capi.STAGE_SHADE); r.synchronize()
q = r.read_queues(); st = r.read_path_state(); qc = r.counters()
n_ext, n_sh = int(qc[7]), int(qc[6])
ext = q[3][:n_ext].copy(); sh = q[4][:n_sh].copy()
ro = O.state_field(st, P, "rayOrigin").view(np.float32)[:, :3]; rd = O.state_field(st, P, "rayDirection").view(np.float32)[:, :3]
so = O.state_field(st, P, "shadowrayOrigin").view(np.float32)[:, :3]
lo = scene["nodes"]["min"][0]; hi = scene["nodes"]["max"][0]; mid = 0.5 * (lo + hi)


def key_of(orig, dirs):
    """eight bins of EQUAL size: median cuts along x, then z, then y of the ray origins (a 3-level kd split) -- or along x only (slabx)"""
    orig = np.nan_to_num(orig)
    n = len(orig)
    key = np.zeros(n, np.int64)
    if key_name == "slabx":
        order = np.argsort(orig[:, 0], kind="stable"); key[order] = (np.arange(n) * 8) // max(n, 1)
        return key
    if key_name == "dir":
        return ((dirs[:, 0] > 0).astype(np.int64) | ((dirs[:, 1] > 0).astype(np.int64) << 1) | ((dirs[:, 2] > 0).astype(np.int64) << 2))
    groups = [np.arange(n)]
    for axis in (0, 2, 1):
        nxt = []
        for gidx in groups:
            o = gidx[np.argsort(orig[gidx, axis], kind="stable")]
            nxt += [o[: len(o) // 2], o[len(o) // 2:]]
        groups = nxt
    for b, gidx in enumerate(groups):
        key[gidx] = b
    return key


def binned(entries, keys):
    """eight bins of one length (a multiple of 128), slot order kept inside a bin, padded with queue holes"""
    valid = entries != 0xFFFFFFFF
    e, k = entries[valid], keys[valid]
    sizes = np.bincount(k, minlength=8)
    L = int((sizes.max() + 127) // 128 * 128)
    out = np.full(8 * L, 0xFFFFFFFF, np.uint32)
    for b in range(8):
        sel = e[k == b]
        out[b * L: b * L + len(sel)] = sel
    return out, L, sizes


ext_keys = np.zeros(n_ext, np.int64); v = ext != 0xFFFFFFFF; ext_keys[v] = key_of(ro[ext[v]], rd[ext[v]])
sh_keys = key_of(so[sh], so[sh])
ext_b, Le, se = binned(ext, ext_keys); sh_b, Ls, ss = binned(sh, sh_keys)
assert 8 * Le <= P and 8 * Ls <= P
print("rays: ext %d shadow %d; bins (%s): ext %s (padded to %d), shadow %s (padded to %d)" % (n_ext, n_sh, key_name, se.tolist(), Le, ss.tolist(), Ls), flush=True)


def time_cast(rr, ext_q, sh_q, reps=9):
    q2 = q.copy(); q2[3][:len(ext_q)] = ext_q; q2[4][:len(sh_q)] = sh_q
    c2 = qc.copy(); c2[7] = len(ext_q); c2[6] = len(sh_q)
    rr.write_path_state(st); rr.write_queues(q2); rr.set_camera(cam.buffer)
    ts = []
    for _ in range(reps):
        rr.write_counters(c2); rr.synchronize(); t0 = time.perf_counter(); rr.run_stage(capi.STAGE_RAYCASTS); rr.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return "min %.3f med %.3f ms" % (min(ts), float(np.median(ts))), rr.read_path_state()


if pmc_mode in (None, "base"):
    t, s_base = time_cast(r, ext, sh); print("queue order as it is            ", t, flush=True)
if pmc_mode in (None, "binned"):
    t, s_b = time_cast(r, ext_b, sh_b); print("binned, XCD rule off            ", t, flush=True)
r.close()
if pmc_mode in (None, "xcd"):
    rx = renderer(True)
    t, s_x = time_cast(rx, ext_b, sh_b); print("binned, a workgroup serves its XCD", t, flush=True)
    rx.close()
if pmc_mode is None:
    f = ["surfacePoint", "baryCoord", "triangle", "isEmitter", "hitDistance", "inShadow"]
    same = all(np.array_equal(O.state_field(s_base, P, n), O.state_field(s_x, P, n)) and np.array_equal(O.state_field(s_base, P, n), O.state_field(s_b, P, n)) for n in f)
    print("results identical in all three:", same)
