"""The multi-GPU path with HIP in every rank: two processes, each with its own renderer (private pool, queues, counters, accumulation
tile) on the device, render config-4 style row bands of the bench scene -- 1920 x 135 at tile_y0 = 135 k, the bands of ranks 3 and 4 of
an 8-rank job over a 1080-row frame -- and exchange the tiles with torch.distributed.gather exactly as bench.py does (gloo here: the test
box has one GPU, so both ranks share device 0; with one GPU per rank the backend is "nccl" = RCCL over xGMI).  Every gathered band must
equal, bit for bit, the oracle run with the same (tile rectangle, pool) parameters.
"""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, WORLD_OF_CONFIG4 = 1920, 1080, 8
RANKS = (3, 4)            # the two bands of the 8-rank split rendered here (the middle of the frame: floor, spheres, walls)
POOL, ITERS = 1 << 16, 6


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _rank_main(rank, world, port, out_path, result_queue):
    try:
        for p in (ROOT, os.path.join(ROOT, "tests")):
            if p not in sys.path:
                sys.path.insert(0, p)
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
        import torch
        import torch.distributed as dist
        import gmupt_pkg
        pkg = gmupt_pkg.load()
        dist.init_process_group("gloo", rank=rank, world_size=world)
        scene = pkg.scenes.build_scene(pkg.scenes.spheres_mesh(202, 3, seed=1234))
        y0, rows = pkg.tiles.row_bands(H, WORLD_OF_CONFIG4)[RANKS[rank]]
        dev = pkg.capi.Device(0)
        sb = pkg.capi.SceneBuffers(dev, scene)
        r = pkg.capi.Renderer(dev, W, rows, pool_paths=POOL, tile=(0, y0))
        r.bind_scene(sb)
        cam = pkg.capi.Camera(W, H); cam.set_pose(*scene["camera"]); cam.buffer.lightCount = scene["light_count"]
        for _ in range(ITERS):
            cam.update(0.0); r.set_camera(cam.buffer); r.iterate()
        tile = torch.empty((rows, W, 4), dtype=torch.float32, device="cuda")
        r.copy_framebuffer_to_device(tile.data_ptr(), tile.numel() * 4)
        # the two bands as a 2-rank frame of 2 * rows rows: the same gather_tiles call as bench.py (CPU tensors for gloo)
        frame = pkg.tiles.gather_tiles(tile.cpu(), W, rows * world, rank, world, dist)
        st = r.stats()
        if rank == 0:
            np.save(out_path, frame.numpy())
        dist.barrier()
        dist.destroy_process_group()
        r.close(); sb.close(); dev.close()
        result_queue.put((rank, "ok", int(st.paths_completed), int(st.flags)))
    except BaseException as e:  # report instead of dying silently: the parent asserts on it
        import traceback
        result_queue.put((rank, "error", traceback.format_exc(), 0))
        raise


@pytest.mark.timeout(600)
def test_two_hip_ranks_render_config4_bands_and_gather(tmp_path, pkg, oracle, clean_process_context):
    ctx = clean_process_context
    world = len(RANKS)
    out = str(tmp_path / "bands.npy")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(k, world, port, out, q)) for k in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=540) for _ in procs]
    for p in procs:
        p.join(60)
    for rank, status, info, flags in results:
        assert status == "ok", "rank %d failed:\n%s" % (rank, info)
        assert flags & (pkg.capi.STAT_CAST_WIDE | pkg.capi.STAT_CAST_FETCH), "rank %d did not run a fused HIP ray-cast kernel" % rank
    assert all(p.exitcode == 0 for p in procs)
    frame = np.load(out)
    bands = pkg.tiles.row_bands(H, WORLD_OF_CONFIG4)
    rows = bands[RANKS[0]][1]
    assert rows == 135 and frame.shape == (rows * world, W, 4)
    scene = pkg.scenes.build_scene(pkg.scenes.spheres_mesh(202, 3, seed=1234))
    for k, rk in enumerate(RANKS):
        y0, n = bands[rk]
        assert (y0, n) == (135 * rk, 135)
        orc = oracle.Renderer(scene, W, n, POOL, tile=(0, y0), threads=8)
        cam = oracle.Camera(W, H); cam.set_pose(*scene["camera"]); cam.buffer.lightCount = scene["light_count"]
        for _ in range(ITERS):
            cam.update(); orc.set_camera(cam.buffer); orc.iterate()
        got = frame[k * rows:(k + 1) * rows]
        assert np.array_equal(got.view(np.uint32), orc.framebuffer().view(np.uint32)), "band of rank %d (rows %d..%d)" % (rk, y0, y0 + n)
        assert int(got[..., 3].view(np.uint32).sum()) == orc.stats().pathsEnded > 0
        orc.close()
