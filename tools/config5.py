"""BASELINE config 5 measured: ~10 M-triangle scene, 3840x2160, pool 2^23, max depth 16 -- the HBM-resident stress configuration.

  python3 tools/config5.py [--steps 40] [--prewarm 60] [--count] [--cache /tmp/config5_scene.npz] [--out FILE] [--spheres 1953]

Prints ONE JSON line: steady-state step time, per-stage HIP-event times, completed paths / s and -- with --count (a deterministic
replay with the counting instantiation of the ray-cast kernel) -- the measured walk statistics (inner nodes I, triangle records T per
ray of either kind, the share of node visits served by the LDS-resident tree top) and from them the records gathered per launch and
per second.  `rocprofv3 --kernel-trace --stats` / `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` are run on this same command (program directly
after `--`; tools/profile_config5.sh); their per-launch bytes divided by the launch time of THIS line is the HBM-side fraction.
--cache keeps the built scene on disk so that the profiler passes of one GPU call do not rebuild it (9 s on 16 cores).
"""
import argparse
import json
import os
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
import gmupt_pkg

ap = argparse.ArgumentParser()
ap.add_argument("--spheres", type=int, default=1953)
ap.add_argument("--subdiv", type=int, default=4)
ap.add_argument("--width", type=int, default=3840)
ap.add_argument("--height", type=int, default=2160)
ap.add_argument("--pool", type=int, default=1 << 23)
ap.add_argument("--max-depth", type=int, default=16)
ap.add_argument("--prewarm", type=int, default=60, help="untimed iterations (max depth 16: the pool is in its steady state after ~3 generations)")
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--count", action="store_true", help="replay with the counting kernels: I, T, LDS share, records per launch")
ap.add_argument("--cache", default="")
ap.add_argument("--soak", type=int, default=0, help="instead of timing: N iterations with the wide ray cast, N with the binary-tree kernel and N with the two separate launches (def0) must leave identical state, counters and framebuffer")
ap.add_argument("--out", default="")
args = ap.parse_args()

g = gmupt_pkg.load(); capi = g.capi
KEYS = ("nodes", "tris", "verts", "props", "lights", "materials", "ref_triangle")
t = time.time()
if args.cache and os.path.exists(args.cache):
    z = np.load(args.cache, allow_pickle=False)
    scene = {k: z[k] for k in KEYS}
    meta = json.loads(str(z["meta"]))
    scene.update(meta); scene["camera"] = tuple(meta["camera"])
    for k in ("tex_diffuse", "tex_metallic_roughness", "tex_normal"):
        scene[k] = None
    build_s = meta["build_s"]
    print("scene from %s in %.1f s" % (args.cache, time.time() - t), file=sys.stderr, flush=True)
else:
    mesh = g.scenes.spheres_mesh(args.spheres, args.subdiv, seed=1234)
    print("generated %d triangles in %.1f s" % (mesh["indices"].shape[0], time.time() - t), file=sys.stderr, flush=True)
    t = time.time(); scene = g.scenes.build_scene(mesh); build_s = time.time() - t
    print("SBVH build + flatten %.1f s: %d nodes, %d references, depth %d, SAH %.1f" % (build_s, scene["nodes"].shape[0], scene["tris"].shape[0], scene["depth"], scene["sah"]), file=sys.stderr, flush=True)
    if args.cache:
        meta = {"light_count": int(scene["light_count"]), "camera": [float(v) for v in scene["camera"]], "name": scene["name"], "sah": float(scene["sah"]),
                "depth": int(scene["depth"]), "num_triangles": int(scene["num_triangles"]), "build_s": round(build_s, 2)}
        np.savez(args.cache, meta=json.dumps(meta), **{k: scene[k] for k in KEYS})

W, H, P = args.width, args.height, args.pool
dev = capi.Device(0)
t = time.time(); sb = capi.SceneBuffers(dev, scene)


def make(stats):
    r = capi.Renderer(dev, W, H, pool_paths=P, tile=(0, 0), max_depth=args.max_depth, collect_stats=stats)
    r.bind_scene(sb)
    cam = capi.Camera(W, H); cam.set_pose(*scene["camera"]); cam.buffer.lightCount = scene["light_count"]
    return r, cam


def step(r, cam, n):
    for _ in range(n):
        cam.update(0.0); r.set_camera(cam.buffer); r.iterate()


if args.soak:
    import hashlib
    digests = {}
    for mode in ("wide", "cast0", "def0"):     # the default kernel, the binary-tree kernel, the two separate launches
        os.environ["GMUPT_TRAVERSAL"] = mode
        r, cam = make(False)
        step(r, cam, args.soak)
        h = hashlib.sha256()
        h.update(np.ascontiguousarray(r.read_path_state()).tobytes()); h.update(r.counters().tobytes()); h.update(np.ascontiguousarray(r.framebuffer()).tobytes())
        st = r.stats()
        digests[mode] = h.hexdigest()
        print(mode, args.soak, "iterations, completed", st.paths_completed, "flags", hex(st.flags), "sha256", digests[mode][:16], flush=True)
        r.close()
    assert len(set(digests.values())) == 1, "the ray-cast kernels differ: %r" % digests
    print("identical")
    sys.exit(0)


r, cam = make(False)
print("upload + bind %.1f s" % (time.time() - t), file=sys.stderr, flush=True)
step(r, cam, args.prewarm); r.synchronize(); r.reset_stats(); r.enable_timing(1)
t = time.perf_counter(); step(r, cam, args.steps); r.synchronize(); dt = time.perf_counter() - t
st = r.stats(); r.enable_timing(0)
n_inner = int((scene["nodes"]["isLeaf"] == 0).sum())
out = {"config": "config5", "triangles": int(scene["num_triangles"]), "nodes": int(scene["nodes"].shape[0]), "references": int(scene["tris"].shape[0]), "depth": int(scene["depth"]),
       "build_s": round(build_s, 1), "width": W, "height": H, "pool": P, "max_depth": args.max_depth, "prewarm": args.prewarm, "steps": args.steps,
       "traversal_bytes": {"node64": n_inner * 64, "tri48": (int(scene["tris"].shape[0]) + 1) * 48},
       "ms_per_step": round(dt / args.steps * 1e3, 4), "mpaths_per_s": round(st.paths_completed / dt / 1e6, 3), "msegments_per_s": round(st.segments / dt / 1e6, 1),
       "stage_ms": {k: round(getattr(st, "ms_" + k) / st.timed_iterations, 4) for k in ("logic", "material", "extend", "shadow")},
       "kernel_flags": {"fused_cast": bool(st.flags & capi.STAT_FUSED_CAST), "k_cast_f": bool(st.flags & capi.STAT_CAST_FETCH), "k_cast_w": bool(st.flags & capi.STAT_CAST_WIDE),
                        "stack_spill_instantiation": bool(st.flags & capi.STAT_STACK_SPILL), "stack_overflow": bool(st.flags & capi.STAT_STACK_OVERFLOW)}}
cast_ms = st.ms_extend / st.timed_iterations
r.close()
if args.count:
    r2, cam2 = make(True)
    step(r2, cam2, args.prewarm); r2.synchronize(); r2.reset_stats()
    step(r2, cam2, args.steps); s2 = r2.stats(); r2.close()
    k = float(args.steps)
    top = s2.ext_top_inner + s2.sh_top_inner
    wide = bool(s2.flags & capi.STAT_CAST_WIDE)
    node_bytes = 112 if wide else 64                            # a WNode is a 128-byte line of which 112 bytes are fetched; a Node64 is 64
    node_recs = (s2.ext_inner + s2.sh_inner - top) / k          # node records fetched from global memory per launch
    tri_recs = (s2.wide_pair_fetches if wide else s2.ext_tris + s2.sh_tris) / k   # triangle records per launch: 48-byte Tri48 (one test), or 80-byte TriPair (one or two tests)
    tri_bytes = 80 if wide else 48
    rays = (s2.ext_rays + s2.sh_rays) / k
    # bytes the kernel's own algorithm moves per launch: every global node visit 64 B, every triangle test 48 B (40 B used), ray in / hit out per ray
    ray_io = (s2.ext_rays * (4 + 24 + 48) + s2.sh_rays * (4 + 28 + 4)) / k
    alg_bytes = node_recs * node_bytes + tri_recs * tri_bytes + ray_io
    # what the REFERENCE's kernels would read for the same walks (144 B per inner step, 52 B per triangle test)
    ref_bytes = (s2.ext_rays * (4 + 24 + 48 + 32 * scene["light_count"] + 48) + 96 * s2.ext_inner + 52 * s2.ext_tris
                 + s2.sh_rays * (4 + 24 + 4 + 48 + 4) + 96 * s2.sh_inner + 52 * s2.sh_tris) / k
    if wide:
        out["traversal_bytes"] = {"wnode128": int(s2.wide_nodes) * 128, "tripair80": int(s2.wide_pairs) * 80, "wide_nodes_in_lds": int(s2.wide_top_nodes), "wide_stack_bound": int(s2.wide_stack_bound)}
    out["cast"] = {"kernel": "k_cast_w" if wide else "k_cast_f", "node_record_bytes_fetched": node_bytes, "redo_rays_per_launch": round(s2.cast_redo_rays / k, 2),
                   "avg_launch_ms": round(cast_ms, 4), "rays_per_launch": rays, "ext_rays_per_launch": s2.ext_rays / k, "shadow_rays_per_launch": s2.sh_rays / k,
                   "inner_per_ext_ray": round(s2.ext_inner / max(s2.ext_rays, 1), 2), "tris_per_ext_ray": round(s2.ext_tris / max(s2.ext_rays, 1), 2),
                   "inner_per_shadow_ray": round(s2.sh_inner / max(s2.sh_rays, 1), 2), "tris_per_shadow_ray": round(s2.sh_tris / max(s2.sh_rays, 1), 2),
                   "lds_top_share_of_node_visits": round(top / max(s2.ext_inner + s2.sh_inner, 1), 4),
                   "global_records_per_launch": node_recs + tri_recs, "grecords_per_s": round((node_recs + tri_recs) / (cast_ms * 1e-3) / 1e9, 2),
                   "kernel_bytes_per_launch": int(alg_bytes), "kernel_bytes_gbs": round(alg_bytes / (cast_ms * 1e-3) / 1e9, 1),
                   "reference_equivalent_bytes_per_launch": None if wide else int(ref_bytes)}   # SURVEY 8(d)'s formula needs the BINARY walk's inner-node count
line = json.dumps(out)
print(line, flush=True)
if args.out:
    with open(args.out, "w") as f:
        f.write(line + "\n")
sb.close(); dev.close()
