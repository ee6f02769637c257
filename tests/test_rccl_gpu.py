"""The RCCL path executed on the GPU box: torch.distributed with backend "nccl" (= RCCL on ROCm) on device tensors.

The box has one GPU, so the group has one rank -- RCCL refuses two ranks on one device -- but every call is the real one:
`init_process_group("nccl", device_id=...)`, a barrier, an all-reduce and `tiles.gather_tiles` through its collective branch
(forced; also with padded bands) on the device tensor the HIP renderer's tile was copied into, exactly as bench.py does with N ranks;
then `bench.py --gpus 1 --force-dist --backend nccl` against the same command without a process group (the deterministic
figures must be equal).  Children come from the fork server started before pytest touched the GPU; in a child torch is imported
FIRST (one HIP runtime per process).
"""
import json
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _nccl_child(port, q):
    try:
        for p in (ROOT, os.path.join(ROOT, "tests")):
            if p not in sys.path:
                sys.path.insert(0, p)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        import gmupt_pkg
        pkg = gmupt_pkg.load()
        capi = pkg.capi
        scene = pkg.scenes.build_scene(pkg.scenes.cornell_mesh())
        W, H, P = 64, 37, 4096      # 37 rows: an odd band
        dev = capi.Device(0)
        sb = capi.SceneBuffers(dev, scene)
        r = capi.Renderer(dev, W, H, pool_paths=P)
        r.bind_scene(sb)
        cam = capi.Camera(W, H); cam.set_pose(*scene["camera"]); cam.buffer.lightCount = scene["light_count"]
        for _ in range(12):
            cam.update(0.0); r.set_camera(cam.buffer); r.iterate()
        tile = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
        r.copy_framebuffer_to_device(tile.data_ptr(), tile.numel() * 4)
        host = r.framebuffer()
        dist.barrier()
        t = torch.tensor([3.0, 4.0], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        frame = pkg.tiles.gather_tiles(tile, W, H, 0, 1, dist, force_collective=True)
        frame_padded = pkg.tiles.gather_tiles(tile, W, H, 0, 1, dist, force_collective=True, pad_rows=H + 5)
        ok = (frame.is_cuda and frame_padded.is_cuda and tuple(frame.shape) == (H, W, 4) and tuple(frame_padded.shape) == (H, W, 4)
              and np.array_equal(frame.cpu().numpy().view(np.uint32), host.view(np.uint32))
              and np.array_equal(frame_padded.cpu().numpy().view(np.uint32), host.view(np.uint32)) and t.tolist() == [3.0, 4.0]
              and int(host[..., 3].view(np.uint32).sum()) > 0)
        # a flagged launch makes the device-to-device hand-over fail as well (GMUPT_ERR_CAST_FAULT)
        os.environ["GMUPT_CAST_LOOP_CAP"] = "2"
        r2 = capi.Renderer(dev, W, H, pool_paths=P); r2.bind_scene(sb)
        del os.environ["GMUPT_CAST_LOOP_CAP"]
        for _ in range(3):
            cam.update(0.0); r2.set_camera(cam.buffer); r2.iterate()
        refused = False
        try:
            r2.copy_framebuffer_to_device(tile.data_ptr(), tile.numel() * 4)
        except capi.GmuptError as e:
            refused = e.code == capi.ERR_CAST_FAULT
        r2.close()
        maps = open("/proc/self/maps").read()
        rccl = sorted({l.split()[-1] for l in maps.splitlines() if "librccl" in l})
        gmupt = sorted({os.path.basename(l.split()[-1]) for l in maps.splitlines() if "libgmupt" in l})
        backend = dist.get_backend()
        dist.destroy_process_group()
        r.close(); sb.close(); dev.close()
        q.put(("ok", ok, refused, rccl, gmupt, backend))
    except BaseException:
        import traceback
        q.put(("error", traceback.format_exc(), False, [], [], ""))
        raise


def _bench_child(port, force_dist, out_path, q):
    try:
        for p in (ROOT, os.path.join(ROOT, "tests")):
            if p not in sys.path:
                sys.path.insert(0, p)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
        import importlib.util
        spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
        bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
        argv = ["bench.py", "--gpus", "1", "--steps", "5", "--warmup", "2", "--prewarm", "30", "--pool", "65536", "--width", "320", "--height", "181", "--spheres", "12", "--subdiv", "2",
                "--spp", "2", "--no-cpu-baseline", "--no-roofline", "--no-config5", "--backend", "nccl"] + (["--force-dist"] if force_dist else [])
        sys.argv = argv
        with open(out_path, "w") as f:
            old = sys.stdout; sys.stdout = f
            try:
                rc = bench.main()
            finally:
                sys.stdout = old
        maps = open("/proc/self/maps").read()
        q.put(("ok", rc, "librccl" in maps))
    except BaseException:
        import traceback
        q.put(("error", traceback.format_exc(), False))
        raise


@pytest.mark.timeout(600)
def test_rccl_gather_on_device_tensors(pkg, clean_process_context):
    ctx = clean_process_context
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_child, args=(_free_port(), q))
    p.start()
    status, ok, refused, rccl, gmupt, backend = q.get(timeout=540)
    p.join(60)
    assert status == "ok", ok
    assert ok, "the gathered frame differs from the renderer's framebuffer"
    assert refused, "a flagged launch must make gmupt_copy_framebuffer_to_device fail"
    assert backend == "nccl" and rccl, "librccl.so must be mapped in the rank's process: %r" % (rccl,)
    assert any(n.startswith("libgmupt") for n in gmupt), gmupt
    assert p.exitcode == 0
    print("rank process mapped", rccl, gmupt)


@pytest.mark.timeout(900)
def test_bench_with_a_one_rank_nccl_group_equals_the_plain_run(tmp_path, clean_process_context):
    ctx = clean_process_context
    lines = {}
    for force in (False, True):
        q = ctx.Queue()
        out = str(tmp_path / ("bench_%d.json" % force))
        p = ctx.Process(target=_bench_child, args=(_free_port(), force, out, q))
        p.start()
        status, rc, rccl = q.get(timeout=800)
        p.join(60)
        assert status == "ok", rc
        assert rc == 0 and p.exitcode == 0
        assert rccl == force or not force, "the forced run must have loaded librccl"
        lines[force] = json.loads(open(out).read().strip().splitlines()[-1])
    a, b = lines[False], lines[True]
    assert a["collectives"].startswith("none") and b["collectives"] == "nccl"
    assert a["n_gpus"] == b["n_gpus"] == 1 and a["metric"] == b["metric"] and a["config"] == b["config"]
    # what does not depend on the clock is equal: the whole-frame job is deterministic
    for k in ("paths", "paths_budget", "iterations", "cut_off_paths"):
        assert a["full_frame"][k] == b["full_frame"][k], k
    assert a["value"] > 0 and b["value"] > 0 and b["tile_gather_ms"] > 0
