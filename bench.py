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

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s tuned float4 copy; 4.6-4.8 TB/s plain grid-stride copy on this pool, profiles/r02_micro/hbm_copy.txt)
# Ceilings of the access pattern that bounds the ray cast: random records gathered by one lane each, 32 waves per CU, dependent chains
# (tools/micro/gather64.hip, measured on this pool: profiles/r02_micro/).  Shape A = 64-byte records as 4 x 16-byte requests (a node),
# shape D = 48-byte records as 3 x 16-byte requests (a triangle); G records/s by table size in MB.  The quad-cooperative shapes (B, E)
# are SLOWER than A on gfx950 and lanes asking for the SAME record (C) are merged -- neither applies to incoherent rays.
GATHER_TABLE_MB = [2.1, 4.2, 8.4, 16.8, 25.6, 1024.0]
GATHER_A_GRECS = [157.4, 153.1, 135.5, 119.3, 105.2, 57.1]
GATHER_D_GRECS = [202.1, 202.4, 192.0, 166.8, 127.6, 45.7]


def gather_ceiling(node_recs, tri_recs, table_mb=None):
    """G records/s of a launch that gathers this mix of node and triangle records at the micro-benchmark's rates: every record an L2 hit
    (table_mb None: the smallest table) or a uniformly random table of table_mb (log-interpolated between the measured sizes)."""
    import math
    if table_mb is None:
        a, d = GATHER_A_GRECS[0], GATHER_D_GRECS[0]
    else:
        x = min(max(table_mb, GATHER_TABLE_MB[0]), GATHER_TABLE_MB[-1])
        k = max(i for i in range(len(GATHER_TABLE_MB) - 1) if GATHER_TABLE_MB[i] <= x)
        t = math.log(x / GATHER_TABLE_MB[k]) / math.log(GATHER_TABLE_MB[k + 1] / GATHER_TABLE_MB[k])
        a = GATHER_A_GRECS[k] + t * (GATHER_A_GRECS[k + 1] - GATHER_A_GRECS[k]); d = GATHER_D_GRECS[k] + t * (GATHER_D_GRECS[k + 1] - GATHER_D_GRECS[k])
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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true", help="skip the counting replay that measures the walk statistics")
    ap.add_argument("--no-full-frame", action="store_true", help="skip the whole-job leg (W*H*spp paths, feed + drain, ~10 s at N = 1)")
    ap.add_argument("--no-stage-timing", action="store_true", help="do not record per-stage HIP events in the timed region")
    ap.add_argument("--cpu-iters", type=int, default=24, help="oracle iterations timed for cpu_baseline (~0.6 s each at pool 2^21 on 16 threads)")
    return ap.parse_args()


def self_launch(args):
    """python bench.py --gpus N outside torchrun: one fresh child per GPU.  Nothing in THIS process initialises a GPU (device_count()
    does not on this image), so starting children is safe; the children get the environment torch.distributed.run would give them."""
    import socket
    import torch
    ndev = torch.cuda.device_count()
    if ndev < args.gpus and args.backend == "nccl":
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible (use --backend gloo to rehearse several ranks on one GPU)" % (args.gpus, ndev))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for rank in range(args.gpus):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    codes = [p.wait() for p in procs]
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        raise SystemExit("bench.py: ranks failed: %r" % (bad,))
    return 0


def roofline_object(capi, s2, steps, cast_ms, scene, table_mb):
    """Roofline of the ray-cast launch from the counting replay (s2) and the HIP-event launch time of the timed region.

    The kernel is NOT bound by HBM on the config-3 scene: its 19 MB of traversal records live in L2 / Infinity Cache.  What bounds it is
    the rate at which the chip gathers random records (tools/micro/gather64.hip), so that is the roof:
      achieved = (64-byte node records + 48-byte triangle records fetched from global memory per launch) / launch time
      peak     = the same mix of records gathered at the micro-benchmark's rate when EVERY record is an L2 hit (2 MB table): no cache
                 behaviour of this per-lane access shape can beat it; `uniform_table` is the rate for a uniformly random table of the
                 size of this scene's traversal records (the walk is not uniform -- the top of the tree is hot -- so a launch may exceed it)
    Node visits served by the LDS-resident tree top issue no vector-memory request and are not counted.  The HBM view is a secondary
    object: `kernel_bytes` are the bytes the kernel's own algorithm moves (64 B per global node visit, 48 B per triangle test, ray in /
    result out), shown against the 8 TB/s HBM peak for orientation only -- most of them are cache hits; measured memory-side traffic
    (rocprofv3 request counters) comes from a profiler run, never from this process: `traffic` is null here and the directory
    of the matching profile is named instead.  `reference_equivalent_bytes` is SURVEY 8(d)'s formula (what the reference's kernels would
    read for the same walks: 144 B per inner step, 52 B per triangle test); it is not a fraction of anything.
    """
    k = float(max(steps, 1))
    fused = bool(s2.flags & capi.STAT_FUSED_CAST)
    inner = s2.ext_inner + (s2.sh_inner if fused else 0)
    top = s2.ext_top_inner + (s2.sh_top_inner if fused else 0)
    tris = s2.ext_tris + (s2.sh_tris if fused else 0)
    node_recs, tri_recs = (inner - top) / k, tris / k
    recs = node_recs + tri_recs
    sec = cast_ms * 1e-3
    achieved = recs / sec / 1e9 if sec > 0 else 0.0
    peak = gather_ceiling(node_recs, tri_recs)
    uniform = gather_ceiling(node_recs, tri_recs, table_mb)
    ray_io = (s2.ext_rays * (4 + 24 + 48) + (s2.sh_rays * (4 + 28 + 4) if fused else 0)) / k
    kernel_bytes = node_recs * 64 + tri_recs * 48 + ray_io
    ref_bytes = (s2.ext_rays * (4 + 24 + 48 + 32 * scene["light_count"] + 48) + 96 * s2.ext_inner + 52 * s2.ext_tris) / k
    if fused:
        ref_bytes += (s2.sh_rays * (4 + 24 + 4 + 48 + 4) + 96 * s2.sh_inner + 52 * s2.sh_tris) / k
    profile = profile_figures("profiles/r02_config3")
    kname = "k_cast_f" if (s2.flags & capi.STAT_CAST_FETCH) else ("fused ray cast (variant)" if fused else "k_extend_d")
    return {"bound": "gather (random 64-byte records; vector-memory request rate)", "kernel": kname,
            "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "Grecords/s", "frac": round(achieved / peak, 4), "traffic": None,
            "peak_source": "tools/micro/gather64.hip shapes A (nodes) and D (triangles), L2-resident table, weighted by this launch's record mix: profiles/r02_micro/",
            "uniform_table": {"table_mb": round(table_mb, 1), "grecords_per_s": round(uniform, 1), "frac": round(achieved / uniform, 4)},
            "avg_launch_ms": round(cast_ms, 4), "records_per_launch": int(recs), "node_records_per_launch": int(node_recs), "triangle_records_per_launch": int(tri_recs),
            "lds_top_share_of_node_visits": round(top / max(inner, 1), 4),
            "rays_per_launch": (s2.ext_rays + (s2.sh_rays if fused else 0)) / k, "shadow_rays_per_launch": s2.sh_rays / k,
            "inner_per_ray": round(s2.ext_inner / max(s2.ext_rays, 1), 2), "tris_per_ray": round(s2.ext_tris / max(s2.ext_rays, 1), 2),
            "shadow_inner_per_ray": round(s2.sh_inner / max(s2.sh_rays, 1), 2), "shadow_tris_per_ray": round(s2.sh_tris / max(s2.sh_rays, 1), 2),
            "simd_efficiency": {"inner": round(s2.ext_inner / max(64 * s2.ext_wave_inner, 1), 3), "triangles": round(s2.ext_tris / max(64 * s2.ext_wave_tris, 1), 3),
                                "shadow_inner": round(s2.sh_inner / max(64 * s2.sh_wave_inner, 1), 3), "shadow_triangles": round(s2.sh_tris / max(64 * s2.sh_wave_tris, 1), 3)},
            "hbm": {"kernel_bytes_per_launch": int(kernel_bytes), "kernel_bytes_gbs": round(kernel_bytes / sec / 1e9, 1) if sec > 0 else 0.0,
                    "hbm_peak_gbs": HBM_PEAK_GBS, "kernel_bytes_over_hbm_peak": round(kernel_bytes / sec / 1e9 / HBM_PEAK_GBS, 4) if sec > 0 else 0.0,
                    "note": "cache hits included: the BVH of this scene is L2 / Infinity-Cache resident; measured FETCH_SIZE / WRITE_SIZE per launch: see traffic_profile",
                    "traffic_profile": "profiles/r02_config3/ (pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes of this command)"},
            "redo_rays_per_launch": round(s2.cast_redo_rays / k, 2),
            "from_profile": profile,
            "reference_equivalent_bytes_per_launch": int(ref_bytes)}


def profile_figures(directory):
    """The counter-based figures of the committed profile of this command (tools/profile_round.sh + tools/roofline_summary.py): memory-side
    bytes (raw FETCH_SIZE, the x2 = 128-byte-request reading, WRITE_SIZE) and the L1 access rate.  They were measured by a separate rocprofv3
    run, NOT by this process -- they are carried with their source so that the line is self-contained, and are None when the file is absent."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), directory, "roofline.json")
    try:
        with open(path) as f:
            j = json.load(f)
    except (OSError, ValueError):
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
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(args.backend)
    coll_dev = "cuda" if (world == 1 or args.backend == "nccl") else "cpu"

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
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def all_sum(vals):
        t = torch.tensor(vals, dtype=torch.float64, device=coll_dev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t.tolist()

    def all_max(val):
        t = torch.tensor([val], dtype=torch.float64, device=coll_dev)
        if world > 1:
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
    frame = tiles.gather_tiles(tile_t if coll_dev == "cuda" else tile_t.cpu(), W, H, rank, world, dist if world > 1 else None)
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
        table_mb = (n_inner * 64 + (int(scene["tris"].shape[0]) + 1) * 48) / 1e6      # the traversal copy: Node64 + Tri48 records
        roofline = roofline_object(capi, s2, args.steps, cast_ms, scene, table_mb)

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
        fr = tiles.gather_tiles(tile_t if coll_dev == "cuda" else tile_t.cpu(), W, H, rank, world, dist if world > 1 else None)
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
            "tile_gather_ms": round(gather_ms, 3), "scene_build_s": round(build_s, 2),
            "full_frame": full_frame, "roofline": roofline, "cpu_baseline": cpu_baseline,
        }
        print(json.dumps(out), flush=True)
    sb.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def run_cpu_baseline(r, cam, scene, W, H, rows, y0, args):
    """The scalar CPU oracle (oracle/, a port of the same six stages) continuing the GPU's own steady-state pool: same scene, resolution,
    camera and POOL SIZE as the timed GPU steps.  The pool (path state, queues, counters, accumulation target) is copied from the device
    into the oracle, one iteration on both sides is compared bit for bit (so what is timed is provably the same computation), then
    args.cpu_iters oracle iterations are timed on the box's host cores."""
    import numpy as np
    import oracle_lib as O
    cores = os.cpu_count() or 1
    threads = min(cores, 16)   # the CPU share of one GPU on this pool is 16 cores
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
    return {"value": round(s.pathsEnded / dt / 1e6, 5), "unit": "Mpaths/s", "cores": threads, "host_cores_visible": cores, "kind": "port",
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
