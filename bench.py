#!/usr/bin/env python3
"""Benchmark of the wavefront path-tracing hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W

With N > 1 and no WORLD_SIZE in the environment this process only LAUNCHES: it starts N fresh child processes (rank i on GPU i,
rendezvous on 127.0.0.1) before anything here has touched a GPU, passes rank 0's output through and exits with the children's status.
Under torch.distributed.run (the driver's way) every rank runs main_rank() directly.

A "step" is one wavefront iteration = one Renderer::draw() of the reference (six stages over the whole path pool,
Source/Renderer.cpp:195-211).  Workload (BASELINE.json configs[2]): the seeded ~260k-triangle scene at 1920x1080, full UE4
PBR + glass + NEE shadow rays, unbounded depth, pool of 2^21 paths per GPU; inputs (scene, path pool) are resident in HBM
before the timed region.  Before the W warm-up steps the pool is pre-warmed to its steady state (paths of all ages in
flight, as during a 64-spp render); `value` is completed camera paths per second over the K timed steps.
Steady state needs care: the reference has no depth limit and kills paths by Russian roulette only after 200 bounces
(logic.hlsl:248-255), so in this closed room ~55 % of the paths end at length 201 and a pool that starts in lock-step
completes paths in bursts with a period of 201 iterations (damping 0.55 per period).  The default pre-warm is ten periods
(2010 iterations, ~3 s) and the default K is one period (201), which makes the value independent of the phase.
With N GPUs the frame is split into N row bands (one private pipeline per rank, no data-path collective); the K timed steps are
bracketed by a barrier and a device synchronisation on both sides.  The RCCL gather of the tiles to rank 0 happens once per frame, not
once per K steps: it is timed on its own right after (`tile_gather_ms`) and is inside the `full_frame` leg (the assembled frame stays on
rank 0's GPU: the metric excludes scene build / upload and the final host read-back).  What is scaled ("weak"): the per-GPU pool is fixed, so every rank does one wavefront
iteration over 2^21 paths per step whatever N is; the frame it fills is 1/N of the image.

Besides `value` the line carries
  full_frame    the WHOLE job of the metric's definition: W*H*spp paths fed, then the drain -- seconds, paths, Mpaths/s;
                with N ranks each renders its band's share (a fixed total job: strong scaling), time = slowest rank
  roofline      for the dominant kernel (k_cast_f, both ray casts in one persistent launch), see roofline_object()
  cpu_baseline  the scalar CPU oracle timed on this box from the GPU's own steady-state pool (rank 0, N = 1 only)
"""
import argparse
import json
import os
import subprocess
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # before torch / HIP are loaded: the host driver only supports dmabuf IPC (RCCL, device tensors across processes)
ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

PROFILE_DIR5 = "profiles/r03_config5"   # the same for the config-5 leg (tools/profile_round.sh <name> config5)
PROFILE_DIR = "profiles/r03_config3"   # the committed rocprofv3 profile of the default command (tools/profile_round.sh)
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s tuned float4 copy; 4.6-4.8 TB/s plain grid-stride copy on this pool, profiles/r02_micro/hbm_copy.txt)
# Ceilings of the access pattern that bounds the ray cast: random records gathered by one lane each, 32 waves per CU, dependent chains
# (tools/micro/gather64.hip).  Shape A = 64-byte records as 4 x 16-byte requests (a binary node), D = 48-byte records as 3 x 16-byte requests
# (a triangle), H = a whole 128-byte line per lane (a wide node); G records/s by table size in MB.  Read from the committed outputs of the
# micro-benchmark (profiles/r02_micro/: measured on this pool in round 2; the box of a later run may differ by a few per cent).
MICRO_DIR = "profiles/r02_micro"


def load_gather_ceilings():
    """{shape: [(table MB, G records/s), ...]} from profiles/r02_micro/gather64_*.txt; None when the files are absent."""
    import re
    tables = {"A": {}, "D": {}, "H": {}}
    base = os.path.join(ROOT, MICRO_DIR)
    try:
        for name in ("gather64_2-16MB.txt", "gather64_25MB.txt", "gather64_1GB.txt"):
            mb = None
            for line in open(os.path.join(base, name)):
                m = re.match(r"table: \d+ records of 64 B = ([0-9.]+) MB", line)
                if m:
                    mb = float(m.group(1)); continue
                m = re.match(r"([AD]) .*? ms\s+([0-9.]+) Grec/s", line)
                if m and mb is not None:
                    tables[m.group(1)][mb] = float(m.group(2))
        for line in open(os.path.join(base, "gather64_shapes_GHI.txt")):
            m = re.match(r"\s+([0-9.]+) MB\s+([0-9.]+)\s+([0-9.]+)\s+([0-9.]+)\s+([0-9.]+)\s*$", line)
            if m:
                tables["H"][float(m.group(1))] = float(m.group(4))
    except OSError:
        return None
    out = {k: sorted(v.items()) for k, v in tables.items()}
    return out if all(out.values()) else None


def _rate(points, table_mb):
    """G records/s at table_mb: the smallest table when None (every record an L2 hit), else log-interpolated between the measured sizes."""
    import math
    if table_mb is None:
        return points[0][1]
    x = min(max(table_mb, points[0][0]), points[-1][0])
    k = max(i for i in range(len(points) - 1) if points[i][0] <= x)
    t = math.log(x / points[k][0]) / math.log(points[k + 1][0] / points[k][0])
    return points[k][1] + t * (points[k + 1][1] - points[k][1])


def gather_ceiling(ceilings, node_recs, tri_recs, node_shape, table_mb=None, tri_shape="D"):
    """G records/s of a launch that gathers this mix of node and triangle records at the micro-benchmark's rates: every record an L2 hit
    (table_mb None) or a uniformly random table of table_mb."""
    a, d = _rate(ceilings[node_shape], table_mb), _rate(ceilings[tri_shape], table_mb)
    return (node_recs + tri_recs) / (node_recs / a + tri_recs / d)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=201)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--prewarm", type=int, default=2010, help="untimed iterations that bring the pool to its steady state (see the module docstring)")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=64, help="samples per pixel of the full_frame leg (the steady-state rate does not depend on it)")
    ap.add_argument("--pool", type=int, default=1 << 21)
    ap.add_argument("--spheres", type=int, default=202)
    ap.add_argument("--subdiv", type=int, default=3)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend: nccl (= RCCL over xGMI) or gloo (rehearsal of N > 1 on one GPU)")
    ap.add_argument("--force-dist", action="store_true", help="N = 1: create the process group anyway and run every collective (barrier, all-reduce, tile gather) through it -- the RCCL path on one GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true", help="skip the counting replay that measures the walk statistics")
    ap.add_argument("--no-full-frame", action="store_true", help="skip the whole-job leg (W*H*spp paths, feed + drain, ~10 s at N = 1)")
    ap.add_argument("--no-stage-timing", action="store_true", help="do not record per-stage HIP events in the timed region")
    ap.add_argument("--no-config5", action="store_true", help="skip the config-5 leg (10 M triangles, 3840x2160, pool 2^23, depth 16: ~25 s incl. the scene build)")
    ap.add_argument("--config5-steps", type=int, default=40)
    ap.add_argument("--config5-prewarm", type=int, default=60)
    ap.add_argument("--cpu-iters", type=int, default=24, help="oracle iterations timed for cpu_baseline (~0.6 s each at pool 2^21 on 16 threads)")
    return ap.parse_args()


def visible_gpus():
    """Number of GPUs this process may use, found WITHOUT touching a GPU or loading HIP: the *_VISIBLE_DEVICES lists if set, else the
    compute nodes of the KFD topology that have SIMDs (CPU nodes have none).  None when it cannot be told."""
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([x for x in v.split(",") if x.strip() != ""])
    top = "/sys/class/kfd/kfd/topology/nodes"
    try:
        n = 0
        for node in os.listdir(top):
            props = dict(l.split()[:2] for l in open(os.path.join(top, node, "properties")) if len(l.split()) >= 2)
            if int(props.get("simd_count", "0")) > 0:
                n += 1
        return n
    except (OSError, ValueError):
        return None


def self_launch(args):
    """python bench.py --gpus N outside torchrun: one fresh child per GPU.  Nothing in THIS process loads torch or HIP (the GPUs are counted
    from sysfs), so starting children is safe; the children get the environment torch.distributed.run would give them and are supervised:
    when one exits with an error the others are terminated instead of waiting in a collective for the c10d timeout."""
    import socket
    ndev = visible_gpus()
    if ndev is not None and ndev < args.gpus and args.backend == "nccl":
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible (use --backend gloo to rehearse several ranks on one GPU)" % (args.gpus, ndev))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for rank in range(args.gpus):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    bad = []
    while procs and not bad:
        time.sleep(0.2)
        for r, pr in enumerate(procs):
            if pr is not None and pr.poll() is not None:
                if pr.returncode != 0:
                    bad.append((r, pr.returncode))
                procs[r] = None
        if all(pr is None for pr in procs):
            break
    if bad:
        for pr in procs:
            if pr is not None:
                pr.terminate()
        for pr in procs:
            if pr is not None:
                try:
                    pr.wait(timeout=20)
                except subprocess.TimeoutExpired:
                    pr.kill()
        raise SystemExit("bench.py: ranks failed: %r (the other ranks were terminated)" % (bad,))
    return 0


def roofline_object(capi, s2, steps, cast_ms, scene, table_mb, profile_dir):
    """Roofline of the ray-cast launch from the counting replay (s2) and the HIP-event launch time of the timed region.

    The kernel is NOT bound by HBM on the config-3 scene: its traversal records (19 MB binary, 17 MB wide) live in L2 / Infinity Cache.  What
    bounds it is the rate at which the chip gathers random records (tools/micro/gather64.hip), so that is the roof:
      achieved = (node records + 48-byte triangle records fetched from global memory per launch) / launch time
                 (a node record is a 64-byte Node64 for k_cast_f, a 128-byte WNode line -- 112 bytes fetched -- for the wide kernel k_cast_w)
      peak     = the same mix of records gathered at the micro-benchmark's rate when EVERY record is an L2 hit (2 MB table: shape A for
                 Node64, shape H for WNode, shape D for triangles): no cache behaviour of this per-lane access shape can beat it;
                 `uniform_table` is the rate for a uniformly random table of the size of this scene's traversal records (the walk is not
                 uniform -- the top of the tree is hot -- so a launch may exceed it)
    Node visits served by the LDS-resident tree top issue no vector-memory request and are not counted.  The HBM view is a secondary
    object: `kernel_bytes` are the bytes the kernel's own algorithm moves (node record bytes per global node visit, 48 B per triangle test,
    ray in / result out), shown against the 8 TB/s HBM peak for orientation only -- most of them are cache hits; measured memory-side
    traffic (rocprofv3 request counters) comes from a profiler run, never from this process: `traffic` is null here and `from_profile`
    carries the committed profile of THIS command (null when this run's workload or kernel is not the profiled one).
    `reference_equivalent_bytes` is SURVEY 8(d)'s formula (what the reference's kernels would read for the same rays: 144 B per inner step
    of the BINARY walk, 52 B per triangle test); it needs the binary walk's inner-node count, so it is only given when k_cast_f ran.
    """
    k = float(max(steps, 1))
    fused = bool(s2.flags & capi.STAT_FUSED_CAST)
    wide = bool(s2.flags & capi.STAT_CAST_WIDE)
    inner = s2.ext_inner + (s2.sh_inner if fused else 0)
    top = s2.ext_top_inner + (s2.sh_top_inner if fused else 0)
    tris = s2.ext_tris + (s2.sh_tris if fused else 0)
    # records fetched from global memory per launch: node records below the LDS-resident top; triangle records -- one 48-byte Tri48 per test
    # (k_cast_f), one 80-byte TriPair per one or two tests (k_cast_w)
    node_recs, tri_recs = (inner - top) / k, (s2.wide_pair_fetches if wide else tris) / k
    recs = node_recs + tri_recs
    sec = cast_ms * 1e-3
    achieved = recs / sec / 1e9 if sec > 0 else 0.0
    ceilings = load_gather_ceilings()
    shape = "H" if wide else "A"
    tri_shape = "A" if wide else "D"      # an 80-byte pair record is five 16-byte requests: priced at the rate of four (shape A), i.e. a little too high a ceiling
    peak = gather_ceiling(ceilings, node_recs, tri_recs, shape, None, tri_shape) if ceilings else None
    uniform = gather_ceiling(ceilings, node_recs, tri_recs, shape, table_mb, tri_shape) if ceilings else None
    node_bytes, tri_bytes = (112, 80) if wide else (64, 48)
    ray_io = (s2.ext_rays * (4 + 24 + 48) + (s2.sh_rays * (4 + 28 + 4) if fused else 0)) / k
    kernel_bytes = node_recs * node_bytes + tri_recs * tri_bytes + ray_io
    ref_bytes = None
    if not wide:
        ref_bytes = (s2.ext_rays * (4 + 24 + 48 + 32 * scene["light_count"] + 48) + 96 * s2.ext_inner + 52 * s2.ext_tris) / k
        if fused:
            ref_bytes += (s2.sh_rays * (4 + 24 + 4 + 48 + 4) + 96 * s2.sh_inner + 52 * s2.sh_tris) / k
    kname = "k_cast_w" if wide else "k_cast_f" if (s2.flags & capi.STAT_CAST_FETCH) else ("fused ray cast (variant)" if fused else "k_extend_d")
    profile = profile_figures(profile_dir, kname) if profile_dir else None
    # memory-side bytes per launch of this kernel from the PMC passes of the committed profile of this command (read requests x 128 B + WRITE_SIZE:
    # the guide's FETCH_SIZE with the gfx950 request-size correction, profiles/r02_micro/fetch_size_calibration.txt); None for any other workload
    traffic = None
    if profile and profile["memory_side_bytes_per_launch"].get("read_128B_requests_x128") is not None:
        traffic = int(profile["memory_side_bytes_per_launch"]["read_128B_requests_x128"] + (profile["memory_side_bytes_per_launch"].get("WRITE_SIZE") or 0))
    return {"bound": "gather (random %s; vector-memory request rate)" % ("128-byte node records + 80-byte triangle-pair records" if wide else "64-byte node records + 48-byte triangle records"), "kernel": kname,
            "achieved": round(achieved, 2), "peak": round(peak, 1) if peak else None, "unit": "Grecords/s", "frac": round(achieved / peak, 4) if peak else None, "traffic": traffic,
            "traffic_unit": "bytes per launch at the L2's memory side (Infinity-Cache hits included), from_profile; this launch's algorithmic bytes: hbm.kernel_bytes_per_launch",
            "peak_source": "tools/micro/gather64.hip shapes %s (nodes) and %s (triangles), L2-resident table, weighted by this launch's record mix: %s/ (measured in round 2 on this pool)" % (shape, tri_shape, MICRO_DIR),
            "uniform_table": {"table_mb": round(table_mb, 1), "grecords_per_s": round(uniform, 1), "frac": round(achieved / uniform, 4)} if uniform else None,
            "avg_launch_ms": round(cast_ms, 4), "records_per_launch": int(recs), "node_records_per_launch": int(node_recs), "triangle_records_per_launch": int(tri_recs), "triangle_tests_per_launch": int(tris / k),
            "lds_top_share_of_node_visits": round(top / max(inner, 1), 4),
            "rays_per_launch": (s2.ext_rays + (s2.sh_rays if fused else 0)) / k, "shadow_rays_per_launch": s2.sh_rays / k,
            "inner_per_ray": round(s2.ext_inner / max(s2.ext_rays, 1), 2), "tris_per_ray": round(s2.ext_tris / max(s2.ext_rays, 1), 2),
            "shadow_inner_per_ray": round(s2.sh_inner / max(s2.sh_rays, 1), 2), "shadow_tris_per_ray": round(s2.sh_tris / max(s2.sh_rays, 1), 2),
            "box_slots_per_node_visit": round(s2.wide_box_tests / max(inner, 1), 2) if wide else 2.0,
            "simd_efficiency": {"inner": round(s2.ext_inner / max(64 * s2.ext_wave_inner, 1), 3), "triangles": round(s2.ext_tris / max(64 * s2.ext_wave_tris, 1), 3),
                                "shadow_inner": round(s2.sh_inner / max(64 * s2.sh_wave_inner, 1), 3), "shadow_triangles": round(s2.sh_tris / max(64 * s2.sh_wave_tris, 1), 3)},
            "redo_rays_per_launch": round(s2.cast_redo_rays / k, 2),
            "general_slab_test_share": round(s2.wide_general_iterations / max(s2.wide_iterations, 1), 4) if wide else None,   # wave iterations with a special 1/d ray walking
            "lane_census": {n: round(v / max(sum(s2.lane_census), 1), 4) for n, v in zip(("no_ray", "walking", "leaf_held_or_no_room", "walk_done_leaves_pending"), s2.lane_census)},
            "helper_subtrees_per_launch": round(s2.cast_helper_subtrees / k, 1),
            "hbm": {"kernel_bytes_per_launch": int(kernel_bytes), "kernel_bytes_gbs": round(kernel_bytes / sec / 1e9, 1) if sec > 0 else 0.0,
                    "hbm_peak_gbs": HBM_PEAK_GBS, "kernel_bytes_over_hbm_peak": round(kernel_bytes / sec / 1e9 / HBM_PEAK_GBS, 4) if sec > 0 else 0.0,
                    "note": ("cache hits included: the traversal records of this scene (%.0f MB) are L2 / Infinity-Cache resident" if table_mb < 200 else "the traversal records of this scene (%.0f MB) exceed the 256 MB Infinity Cache: most of these bytes come from HBM") % table_mb + "; measured memory-side bytes per launch: from_profile"},
            "from_profile": profile,
            "reference_equivalent_bytes_per_launch": int(ref_bytes) if ref_bytes is not None else None}


def profile_figures(directory, kernel):
    """The counter-based figures of the committed profile of this command (tools/profile_round.sh + tools/roofline_summary.py): memory-side
    bytes (raw FETCH_SIZE, the x2 = 128-byte-request reading, WRITE_SIZE) and the L1 access rate.  They were measured by a separate rocprofv3
    run, NOT by this process -- they are carried with their source so that the line is self-contained; None when the file is absent or
    describes another kernel."""
    path = os.path.join(ROOT, directory, "roofline.json")
    try:
        with open(path) as f:
            j = json.load(f)
    except (OSError, ValueError):
        return None
    if j.get("kernel") != kernel:
        return None
    m = j.get("memory_side", {})
    ms = j.get("avg_launch_ms_rocprof_trace") or 0.0
    raw = m.get("FETCH_SIZE_bytes_raw", 0) + m.get("write_bytes", 0)
    return {"source": directory + "/roofline.json (a separate rocprofv3 run of this command; not measured by this process)",
            "avg_launch_ms_rocprof_trace": ms,
            "memory_side_bytes_per_launch": {"FETCH_SIZE_raw": m.get("FETCH_SIZE_bytes_raw"), "read_128B_requests_x128": m.get("read_bytes"), "WRITE_SIZE": m.get("write_bytes")},
            "memory_side_gbs": {"raw_FETCH_SIZE": round(raw / (ms * 1e-3) / 1e9, 1) if ms else None, "x2": m.get("gbs")},
            "memory_side_over_hbm_peak": {"raw_FETCH_SIZE": round(raw / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if ms else None, "x2": m.get("frac_of_hbm_peak")},
            "note": "Infinity-Cache (MALL) hits are inside these counters; every memory-side read request of this kernel is a 128-byte line, FETCH_SIZE tallies it at 64 bytes (profiles/r02_micro/fetch_size_calibration.txt)",
            "vmem_request_rate": j.get("vmem_request_rate")}


def main_rank(args):
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE %d != --gpus %d" % (world, args.gpus))

    import torch  # first: the process must use ONE HIP runtime (torch's bundled libamdhip64.so.7, same SONAME as ROCm's)
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False)")
    ndev = torch.cuda.device_count()
    if world > ndev and args.backend == "nccl":
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible (use --backend gloo to rehearse several ranks on one GPU)" % (world, ndev))
    dev_index = local_rank % ndev      # one rank per GPU; the modulo only matters for the gloo rehearsal on a single GPU
    torch.cuda.set_device(dev_index)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(args.backend)
    coll_dev = "cuda" if (not use_dist or args.backend == "nccl") else "cpu"

    import numpy as np
    import gmupt_pkg
    pkg = gmupt_pkg.load()
    capi, scenes, tiles = pkg.capi, pkg.scenes, pkg.tiles

    W, H = args.width, args.height
    t0 = time.time()
    scene = scenes.build_scene(scenes.spheres_mesh(args.spheres, args.subdiv, seed=1234))
    build_s = time.time() - t0
    bands = tiles.row_bands(H, world)
    y0, rows = bands[rank]

    dev = capi.Device(dev_index)
    sb = capi.SceneBuffers(dev, scene)

    def make_renderer(stats, budget=0):
        r = capi.Renderer(dev, W, rows, pool_paths=args.pool, tile=(0, y0), collect_stats=stats, path_budget=budget)
        r.bind_scene(sb)
        cam = capi.Camera(W, H)
        cam.set_pose(*scene["camera"])
        cam.buffer.lightCount = scene["light_count"]
        return r, cam

    def step(r, cam, n):
        for _ in range(n):
            cam.update(0.0)            # Renderer::update: new randomSeed pair, iterationCounter++
            r.set_camera(cam.buffer)   # 112-byte camera upload
            r.iterate()                # Renderer::draw: the six stages

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def all_sum(vals):
        t = torch.tensor(vals, dtype=torch.float64, device=coll_dev)
        if use_dist:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t.tolist()

    def all_max(val):
        t = torch.tensor([val], dtype=torch.float64, device=coll_dev)
        if use_dist:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # ------------------------------------------------------------------ the timed K steps (steady state)
    r, cam = make_renderer(False)
    step(r, cam, args.prewarm + args.warmup)
    r.synchronize()
    r.reset_stats()
    r.enable_timing(0 if args.no_stage_timing else 2)   # two HIP events per step, around the ray-cast launch only
    tile_t = torch.empty((rows, W, 4), dtype=torch.float32, device="cuda")

    barrier()
    t_start = time.perf_counter()
    step(r, cam, args.steps)
    r.synchronize()
    barrier()
    elapsed = time.perf_counter() - t_start
    # the tile gather happens once per FRAME (thousands of steps), not once per K steps: it is timed on its own here and is part of the
    # full_frame leg below, where it belongs
    t_gather = time.perf_counter()
    r.copy_framebuffer_to_device(tile_t.data_ptr(), tile_t.numel() * 4)  # synchronises the renderer's stream
    frame = tiles.gather_tiles(tile_t if coll_dev == "cuda" else tile_t.cpu(), W, H, rank, world, dist if use_dist else None, force_collective=args.force_dist)
    barrier()
    gather_ms = (time.perf_counter() - t_gather) * 1e3

    st = r.stats()
    if st.flags & (capi.STAT_STACK_OVERFLOW | capi.STAT_CAST_ABORTED):
        raise SystemExit("bench.py: the ray cast reported an error (gmupt_stats.flags = %#x): the numbers would be invalid" % st.flags)
    r.enable_timing(0)
    cast_ms = st.ms_extend / max(st.timed_iterations, 1)
    # per-stage breakdown: a short untimed continuation with events around every stage group
    r.reset_stats(); r.enable_timing(1); step(r, cam, 50); stb = r.stats(); r.enable_timing(0)
    total_paths, total_segments = all_sum([float(st.paths_completed), float(st.segments)])
    elapsed = all_max(elapsed)
    gather_ms = all_max(gather_ms)

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_baseline = run_cpu_baseline(r, cam, scene, W, H, rows, y0, args)
    r.close()
    del frame

    # ------------------------------------------------------------------ roofline: deterministic replay with the counting kernels
    roofline = None
    if rank == 0 and not args.no_roofline:
        r2, cam2 = make_renderer(True)
        step(r2, cam2, args.prewarm + args.warmup)
        r2.synchronize(); r2.reset_stats()
        step(r2, cam2, args.steps)
        s2 = r2.stats()
        r2.close()
        n_inner = int((scene["nodes"]["isLeaf"] == 0).sum())
        wide = bool(s2.flags & capi.STAT_CAST_WIDE)
        node_table = s2.wide_nodes * 128 if wide else n_inner * 64
        table_mb = (node_table + (s2.wide_pairs * 80 if wide else (int(scene["tris"].shape[0]) + 1) * 48)) / 1e6      # the traversal copy the kernel walked: WNode + TriPair, or Node64 + Tri48 records
        # the committed profile belongs to the default workload only
        default_workload = (args.width, args.height, args.pool, args.spheres, args.subdiv) == (1920, 1080, 1 << 21, 202, 3)
        roofline = roofline_object(capi, s2, args.steps, cast_ms, scene, table_mb, PROFILE_DIR if default_workload else None)

    # ------------------------------------------------------------------ the whole job: W*H*spp paths, feed + drain (each rank its band)
    full_frame = None
    if not args.no_full_frame:
        budget = W * rows * args.spp
        rf, camf = make_renderer(False, budget=budget)
        rf.synchronize()
        barrier()
        t1 = time.perf_counter()
        iters = rf.render_budget(camf)                       # Camera::update + upload + iterate per frame, until the budget has drained
        rf.copy_framebuffer_to_device(tile_t.data_ptr(), tile_t.numel() * 4)
        fr = tiles.gather_tiles(tile_t if coll_dev == "cuda" else tile_t.cpu(), W, H, rank, world, dist if use_dist else None, force_collective=args.force_dist)
        barrier()
        ff_s = time.perf_counter() - t1
        sf = rf.stats()
        if sf.flags & (capi.STAT_STACK_OVERFLOW | capi.STAT_CAST_ABORTED):
            raise SystemExit("bench.py: the ray cast reported an error in the full-frame leg (gmupt_stats.flags = %#x)" % sf.flags)
        done, cut = all_sum([float(sf.paths_completed), float(sf.active_paths)])
        ff_s = all_max(ff_s); iters = int(all_max(float(iters)))
        rf.close(); del fr
        full_frame = {"seconds": round(ff_s, 3), "paths": int(done), "paths_budget": W * H * args.spp, "iterations": iters, "mpaths_per_s": round(done / ff_s / 1e6, 3),
                      "cut_off_paths": int(cut), "spp": args.spp, "scaling": "strong" if world > 1 else "n/a",
                      "note": "feed W*H*spp paths then drain; the drain ends 512 iterations after the budget ran out (the reference's NaN-throughput paths never end, DESIGN.md section 5)"}

    config5 = None
    if rank == 0 and world == 1 and not args.no_config5:
        config5 = config5_leg(capi, scenes, dev, args)

    if rank == 0:
        value = total_paths / elapsed / 1e6
        out = {
            "metric": "Mpaths/s @1920x1080x64spp, 260k-tri scene", "value": round(value, 3), "unit": "Mpaths/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "headline": "value = completed paths / wall time of the K timed steady-state steps (all ranks; barrier + device synchronisation on both sides); tile_gather_ms = one gather of the tiles to rank 0, which happens once per frame and is inside full_frame = the whole W*H*spp job incl. ramp-up, drain and gather",
            "config": {"workload": "config3: %d-tri seeded sphere room (%d nodes), %dx%d, %d spp steady state, UE4+glass+NEE, unbounded depth"
                       % (scene["num_triangles"], scene["nodes"].shape[0], W, H, args.spp),
                       "pool_paths_per_gpu": args.pool, "prewarm_steps": args.prewarm, "tiling": "row bands x%d" % world,
                       "scaled": "per-GPU pool fixed; a step = one wavefront iteration of every rank's pool; the frame is split into %d row band(s)" % world,
                       "parallelism": "tile%d" % world},
            "msegments_per_s": round(total_segments / elapsed / 1e6, 1),
            "stage_ms": {"logic": round(stb.ms_logic / max(stb.timed_iterations, 1), 4), "material": round(stb.ms_material / max(stb.timed_iterations, 1), 4),
                         "raycast": round(cast_ms, 4), "shadow_separate": round(stb.ms_shadow / max(stb.timed_iterations, 1), 4)},
            "tile_gather_ms": round(gather_ms, 3), "scene_build_s": round(build_s, 2), "collectives": (args.backend if use_dist else "none (one rank)"),
            "full_frame": full_frame, "roofline": roofline, "cpu_baseline": cpu_baseline, "config5": config5,
        }
        print(json.dumps(out), flush=True)
    sb.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def config5_leg(capi, scenes, dev, args):
    """BASELINE config 5 on this box (rank 0, N = 1): the seeded 10 M-triangle scene (1 953 icospheres at subdivision 4 + room), 3840x2160,
    pool 2^23, max depth 16 -- the one configuration whose traversal records (0.7 GB) do not fit the 256 MB Infinity Cache, i.e. where
    north_star's "traverse kernel vs HBM roofline" is judged.  Steady-state step time, the ray-cast launch time (HIP events), the walk
    statistics of a counting replay and from them the records gathered and the bytes the kernel's algorithm moves per launch; the
    memory-side bytes of the committed rocprofv3 profile of the same workload are carried as `from_profile`."""
    import numpy as np
    t0 = time.time()
    scene = scenes.build_scene(scenes.spheres_mesh(1953, 4, seed=1234))
    build_s = time.time() - t0
    W, H, P, depth = 3840, 2160, 1 << 23, 16
    t0 = time.time()
    sb = capi.SceneBuffers(dev, scene)

    def make(stats):
        r = capi.Renderer(dev, W, H, pool_paths=P, tile=(0, 0), max_depth=depth, collect_stats=stats)
        r.bind_scene(sb)
        cam = capi.Camera(W, H); cam.set_pose(*scene["camera"]); cam.buffer.lightCount = scene["light_count"]
        return r, cam

    def step(r, cam, n):
        for _ in range(n):
            cam.update(0.0); r.set_camera(cam.buffer); r.iterate()

    r, cam = make(False)
    bind_s = time.time() - t0
    step(r, cam, args.config5_prewarm); r.synchronize(); r.reset_stats(); r.enable_timing(1)
    t0 = time.perf_counter(); step(r, cam, args.config5_steps); r.synchronize(); dt = time.perf_counter() - t0
    st = r.stats(); r.enable_timing(0)
    cast_ms = st.ms_extend / max(st.timed_iterations, 1)
    r.close()
    r2, cam2 = make(True)
    step(r2, cam2, args.config5_prewarm); r2.synchronize(); r2.reset_stats()
    step(r2, cam2, args.config5_steps); s2 = r2.stats(); r2.close()
    n_inner = int((scene["nodes"]["isLeaf"] == 0).sum())
    wide = bool(s2.flags & capi.STAT_CAST_WIDE)
    table_mb = (s2.wide_nodes * 128 + s2.wide_pairs * 80 if wide else n_inner * 64 + (int(scene["tris"].shape[0]) + 1) * 48) / 1e6
    roof = roofline_object(capi, s2, args.config5_steps, cast_ms, scene, table_mb, PROFILE_DIR5)
    sb.close()
    prof = roof.get("from_profile") or {}
    return {"workload": "config5: %d-tri seeded sphere room (%d nodes, %d references, depth %d), %dx%d, pool 2^23, max depth %d"
                        % (scene["num_triangles"], scene["nodes"].shape[0], scene["tris"].shape[0], scene["depth"], W, H, depth),
            "steps": args.config5_steps, "prewarm": args.config5_prewarm, "scene_build_s": round(build_s, 1), "upload_bind_s": round(bind_s, 1),
            "ms_per_step": round(dt / args.config5_steps * 1e3, 4), "mpaths_per_s": round(st.paths_completed / dt / 1e6, 3), "msegments_per_s": round(st.segments / dt / 1e6, 1),
            "stage_ms": {"logic": round(st.ms_logic / max(st.timed_iterations, 1), 4), "material": round(st.ms_material / max(st.timed_iterations, 1), 4), "raycast": round(cast_ms, 4)},
            "kernel_flags": {"fused_cast": bool(st.flags & capi.STAT_FUSED_CAST), "k_cast_f": bool(st.flags & capi.STAT_CAST_FETCH), "k_cast_w": bool(st.flags & capi.STAT_CAST_WIDE),
                             "stack_spill_instantiation": bool(st.flags & capi.STAT_STACK_SPILL)},
            "traversal_table_mb": round(table_mb, 1),
            "hbm_fraction_from_profile": (prof.get("memory_side_over_hbm_peak") or {}).get("x2"),
            "roofline": roof}


def run_cpu_baseline(r, cam, scene, W, H, rows, y0, args):
    """The scalar CPU oracle (oracle/, a port of the same six stages) continuing the GPU's own steady-state pool: same scene, resolution,
    camera and POOL SIZE as the timed GPU steps.  The pool (path state, queues, counters, accumulation target) is copied from the device
    into the oracle, one iteration on both sides is compared bit for bit (so what is timed is provably the same computation), then
    args.cpu_iters oracle iterations are timed on the box's host cores."""
    import numpy as np
    import oracle_lib as O
    cores = os.cpu_count() or 1
    threads = min(cores, 16)   # the CPU share of one GPU on this pool is 16 cores (the box shows all 256 of the host, but a job may only load 16)
    orc = O.Renderer(scene, W, rows, args.pool, tile=(0, y0), threads=threads)
    orc.path_state()[:] = r.read_path_state(); orc.queues()[:] = r.read_queues(); orc.counters()[:] = r.counters(); orc.framebuffer()[:] = r.framebuffer()

    def both(n):
        for _ in range(n):
            cam.update(0.0); r.set_camera(cam.buffer); r.iterate(); orc.set_camera(cam.buffer); orc.iterate()
    both(1)
    import parity_util as PU
    verified = bool(np.array_equal(orc.counters(), r.counters()) and not PU.compare_state(orc, r, args.pool, args.pool)
                    and np.array_equal(orc.framebuffer().view(np.uint32), r.framebuffer().view(np.uint32)))
    orc.reset_stats()
    t1 = time.perf_counter()
    for _ in range(args.cpu_iters):
        cam.update(0.0); orc.set_camera(cam.buffer); orc.iterate()
    dt = time.perf_counter() - t1
    s = orc.stats()
    orc.close()
    return {"value": round(s.pathsEnded / dt / 1e6, 5), "unit": "Mpaths/s", "cores": threads, "host_cores_visible": cores, "cores_note": "16-core share of one GPU on this pool (the host's other cores belong to the other GPUs' jobs)", "kind": "port",
            "verified_against_gpu": verified,
            "sample": "oracle (scalar C port of the six stages) continuing the GPU's steady-state pool: pool %d (= the GPU's), %d timed iterations (%.1f s), same scene / resolution / camera; all stages on %d OpenMP threads (per-slot work parallel, queue / framebuffer order applied serially)"
                      % (args.pool, args.cpu_iters, dt, threads),
            "msegments_per_s": round(s.segments / dt / 1e6, 3)}


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args)
    return main_rank(args)


if __name__ == "__main__":
    sys.exit(main())
