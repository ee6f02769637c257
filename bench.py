#!/usr/bin/env python3
"""Benchmark of the wavefront path-tracing hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" is one wavefront iteration = one Renderer::draw() of the reference (six stages over the whole path pool,
Source/Renderer.cpp:195-211).  Workload (BASELINE.json configs[2]): the seeded ~260k-triangle scene at 1920x1080, full UE4
PBR + glass + NEE shadow rays, unbounded depth, pool of 2^21 paths per GPU; inputs (scene, path pool) are resident in HBM
before the timed region.  Before the W warm-up steps the pool is pre-warmed to its steady state (paths of all ages in
flight, as during a 64-spp render); the value is completed camera paths per second over the K timed steps.
Steady state needs care: the reference has no depth limit and kills paths by Russian roulette only after 200 bounces
(logic.hlsl:248-255), so in this closed room ~55 % of the paths end at length 201 and a pool that starts in lock-step
completes paths in bursts with a period of 201 iterations (damping 0.55 per period).  The default pre-warm is ten periods
(2010 iterations, ~4 s) and the default K is one period (201), which makes the value independent of the phase.
With N GPUs the frame is split into N row bands (one private pipeline per rank, no data-path collective); the timed region
ends with the RCCL gather of the tiles to rank 0 (the assembled frame stays on rank 0's GPU: the metric excludes scene build / upload
and the final host read-back).

Prints ONE JSON line on rank 0 (see the task contract); `roofline` is for the dominant kernel (the ray-cast launch: extension + shadow rays in one persistent kernel) and
`cpu_baseline` is the scalar CPU oracle timed on a bounded sample on this box (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def ext_bytes(rays, inner, tris, lights):
    """Algorithmic bytes of the extension ray cast: 4 + 24 + 48(1 + 2I) + 52T + 32L + 48 per ray."""
    return rays * (4 + 24 + 48 + 32 * lights + 48) + 96 * inner + 52 * tris


def shadow_bytes(rays, inner, tris):
    """Algorithmic bytes of the shadow ray cast: 4 + 24 + 4 + 48(1 + 2I) + 52T + 4 per ray."""
    return rays * (4 + 24 + 4 + 48 + 4) + 96 * inner + 52 * tris


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=201)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--prewarm", type=int, default=2010, help="untimed iterations that bring the pool to its steady state (see the module docstring)")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=64, help="named in the config; the steady-state rate does not depend on it")
    ap.add_argument("--pool", type=int, default=1 << 21)
    ap.add_argument("--spheres", type=int, default=202)
    ap.add_argument("--subdiv", type=int, default=3)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend: nccl (= RCCL over xGMI) or gloo (rehearsal of N > 1 on one GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true", help="skip the counting replay that measures I and T")
    ap.add_argument("--no-stage-timing", action="store_true", help="do not record per-stage HIP events in the timed region")
    ap.add_argument("--cpu-pool", type=int, default=1 << 17)
    ap.add_argument("--cpu-iters", type=int, default=201)
    ap.add_argument("--cpu-prewarm", type=int, default=402)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE %d != --gpus %d" % (world, args.gpus))

    import torch  # first: the process must use ONE HIP runtime (torch's bundled libamdhip64.so.7, same SONAME as ROCm's)
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False)")
    ndev = torch.cuda.device_count()
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

    def make_renderer(stats):
        r = capi.Renderer(dev, W, rows, pool_paths=args.pool, tile=(0, y0), collect_stats=stats)
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

    r, cam = make_renderer(False)
    step(r, cam, args.prewarm + args.warmup)
    r.synchronize()
    r.reset_stats()
    r.enable_timing(0 if args.no_stage_timing else 2)   # two HIP events per step, around the extension ray cast only
    tile_t = torch.empty((rows, W, 4), dtype=torch.float32, device="cuda")

    barrier()
    t_start = time.perf_counter()
    step(r, cam, args.steps)
    r.copy_framebuffer_to_device(tile_t.data_ptr(), tile_t.numel() * 4)  # synchronises the renderer's stream
    frame = tiles.gather_tiles(tile_t if coll_dev == "cuda" else tile_t.cpu(), W, H, rank, world, dist if world > 1 else None)
    barrier()
    elapsed = time.perf_counter() - t_start

    st = r.stats()
    r.enable_timing(0)
    ext_ms = st.ms_extend / max(st.timed_iterations, 1)
    # per-stage breakdown: a short untimed continuation with events around every stage group
    r.reset_stats(); r.enable_timing(1); step(r, cam, 50); stb = r.stats(); r.enable_timing(0)
    completed = torch.tensor([float(st.paths_completed), float(st.segments)], dtype=torch.float64, device=coll_dev)
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
    if world > 1:
        dist.all_reduce(completed, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    total_paths, total_segments = completed.tolist()
    elapsed = float(tmax.item())

    roofline = None
    if rank == 0 and not args.no_roofline:
        # replay the same (deterministic) sequence with the counting variant of the traverse kernels to get the
        # measured I (inner nodes visited) and T (triangle references tested) of the timed extension launches
        r2, cam2 = make_renderer(True)
        step(r2, cam2, args.prewarm + args.warmup)
        r2.synchronize(); r2.reset_stats()
        step(r2, cam2, args.steps)
        s2 = r2.stats()
        r2.close()
        nbytes = ext_bytes(s2.ext_rays, s2.ext_inner, s2.ext_tris, scene["light_count"])
        fused = bool(s2.flags & capi.STAT_FUSED_CAST)   # one launch casts the extension AND the shadow rays: its bytes are the sum
        if fused:
            nbytes += shadow_bytes(s2.sh_rays, s2.sh_inner, s2.sh_tris)
        kname = {"cast0": "k_cast_f", "cast3": "k_cast_f", "cast2": "k_cast_m", "cast1": "k_cast_d"}.get(os.environ.get("GMUPT_TRAVERSAL", "cast0"), "k_cast_f") if fused else "k_extend_d"
        per_launch = nbytes / max(args.steps, 1)
        achieved = per_launch / (ext_ms * 1e-3) / 1e9 if ext_ms > 0 else 0.0
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get(kname + "_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "algorithmic_bytes_per_launch": int(per_launch), "avg_launch_ms": round(ext_ms, 4),
                    "rays_per_launch": (s2.ext_rays + (s2.sh_rays if fused else 0)) / max(args.steps, 1),
                    "shadow_rays_per_launch": s2.sh_rays / max(args.steps, 1), "shadow_inner_per_ray": round(s2.sh_inner / max(s2.sh_rays, 1), 2),
                    "shadow_tris_per_ray": round(s2.sh_tris / max(s2.sh_rays, 1), 2),
                    "inner_per_ray": round(s2.ext_inner / max(s2.ext_rays, 1), 2), "tris_per_ray": round(s2.ext_tris / max(s2.ext_rays, 1), 2),
                    "simd_efficiency": {"inner": round(s2.ext_inner / max(64 * s2.ext_wave_inner, 1), 3), "triangles": round(s2.ext_tris / max(64 * s2.ext_wave_tris, 1), 3),
                                        "shadow_inner": round(s2.sh_inner / max(64 * s2.sh_wave_inner, 1), 3), "shadow_triangles": round(s2.sh_tris / max(64 * s2.sh_wave_tris, 1), 3)}}

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_baseline = run_cpu_baseline(scene, W, H, args)

    if rank == 0:
        value = total_paths / elapsed / 1e6
        out = {
            "metric": "Mpaths/s @1920x1080x64spp, 260k-tri scene", "value": round(value, 3), "unit": "Mpaths/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "config3: %d-tri seeded sphere room (%d nodes), %dx%d, %d spp steady state, UE4+glass+NEE, unbounded depth"
                       % (scene["num_triangles"], scene["nodes"].shape[0], W, H, args.spp),
                       "pool_paths_per_gpu": args.pool, "prewarm_steps": args.prewarm, "tiling": "row bands x%d" % world,
                       "parallelism": "tile%d" % world},
            "msegments_per_s": round(total_segments / elapsed / 1e6, 1),
            "stage_ms": {"logic": round(stb.ms_logic / max(stb.timed_iterations, 1), 4), "scan": round(stb.ms_scan / max(stb.timed_iterations, 1), 4),
                         "material": round(stb.ms_material / max(stb.timed_iterations, 1), 4), "extend": round(ext_ms, 4),
                         "shadow": round(stb.ms_shadow / max(stb.timed_iterations, 1), 4)},
            "scene_build_s": round(build_s, 2),
            "roofline": roofline, "cpu_baseline": cpu_baseline,
        }
        print(json.dumps(out), flush=True)
    r.close()
    sb.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def run_cpu_baseline(scene, W, H, args):
    """The scalar CPU oracle (oracle/, a port of the same six stages) on a bounded sample of the same workload."""
    import oracle_lib as O
    threads = min(os.cpu_count() or 1, 16)   # the box's CPU share for one GPU
    pool = args.cpu_pool
    orc = O.Renderer(scene, W, H, pool, threads=threads)
    cam = O.Camera(W, H)
    cam.set_pose(*scene["camera"])
    cam.buffer.lightCount = scene["light_count"]

    def step(n):
        for _ in range(n):
            cam.update(); orc.set_camera(cam.buffer); orc.iterate()
    t0 = time.perf_counter()
    step(args.cpu_prewarm)   # periods of the 201-iteration completion cycle (see the module docstring)
    orc.reset_stats()
    t1 = time.perf_counter()
    step(args.cpu_iters)
    dt = time.perf_counter() - t1
    s = orc.stats()
    orc.close()
    return {"value": round(s.pathsEnded / dt / 1e6, 5), "unit": "Mpaths/s", "cores": threads, "kind": "port",
            "sample": "oracle (scalar C port of the six stages), pool %d, %d timed iterations = one completion period (%.1f s) after %d pre-warm iterations (%.1f s), same scene / resolution / camera; all stages on %d OpenMP threads (per-slot work parallel, queue / framebuffer order applied serially)"
                      % (pool, args.cpu_iters, dt, args.cpu_prewarm, t1 - t0, threads),
            "msegments_per_s": round(s.segments / dt / 1e6, 3)}


if __name__ == "__main__":
    main()
