"""f4: the interactive rules of the reference without a window -- camera input with the accumulation-reset hysteresis
(Source/Camera.cpp:25-83), and the headless progressive session that gathers tiles periodically (gloo, 2 ranks; the ranks render
with the CPU oracle because this is a CPU test, the session code is the product's)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_camera_input_and_reset_hysteresis(pkg):
    cam = pkg.capi.Camera(64, 36)
    cam.set_pose(1.0, 3.0, 8.0, 0.0, 270.0)
    counter, hyst, expect = -1, False, []
    # script: (frames, input?)  -- idle, a mouse drag, idle again, a key held, idle
    script = [(8, None), (3, "mouse"), (9, None), (4, "key"), (8, None)]
    got = []
    for frames, what in script:
        for _ in range(frames):
            if what == "mouse":
                cam.set_input(mouse_dx=1.5, mouse_dy=-0.5)
            elif what == "key":
                cam.set_input(w=True)
            else:
                cam.set_input()
            cam.update(0.01)
            got.append(int(cam.buffer.iterationCounter))
            # Camera.cpp:70-83
            counter += 1
            if what is not None:
                if counter > 4:
                    counter = 0
                hyst = True
            elif hyst and counter > 4:
                counter = 0
                hyst = False
            expect.append(counter)
    assert got == expect
    assert 0 in got[8:11] and got[7] == 7           # the drag restarted a frame that had accumulated more than 4 iterations
    # the drag turned the camera (3 x 1.5 degrees of yaw, 3 x 0.5 of pitch) and W moved it 4 x 5 x 0.01 along the view direction
    yaw, pitch = np.deg2rad(270.0 + 4.5), np.deg2rad(1.5)
    front = np.array([np.cos(yaw) * np.cos(pitch), np.sin(pitch), np.sin(yaw) * np.cos(pitch)])
    pos = np.array(list(cam.buffer.position)[:3])
    assert np.allclose(pos, np.array([1.0, 3.0, 8.0]) + front * 0.2, atol=1e-5)
    horiz = np.array(list(cam.buffer.horizontal)[:3])
    assert abs(np.dot(horiz, front)) < 1e-5 and abs(horiz[1]) < 1e-6
    # pitch is clamped to +-89 degrees (Camera.cpp:32-33)
    cam.set_input(mouse_dy=-500.0); cam.update(0.0)
    ulc = np.array(list(cam.buffer.upperLeftCorner)[:3]); h = np.array(list(cam.buffer.horizontal)[:3]); v = np.array(list(cam.buffer.vertical)[:3])
    centre = ulc + 0.5 * h - 0.5 * v
    assert abs(np.degrees(np.arcsin(centre[1] / np.linalg.norm(centre))) - 89.0) < 1e-3


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, W, H, P, out_path):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    import gmupt_pkg
    import oracle_lib as O
    pkg = gmupt_pkg.load()
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene = pkg.scenes.build_scene(pkg.scenes.cornell_mesh())
    y0, rows = pkg.tiles.row_bands(H, world)[rank]
    orc = O.Renderer(scene, W, rows, P, tile=(0, y0))
    cam = pkg.capi.Camera(W, H); cam.set_pose(*scene["camera"])
    seen = []
    sess = pkg.progressive.ProgressiveSession(orc, cam, W, H, rank, world, dist, preview_every=6,
                                              on_preview=lambda n, f: seen.append((n, f.copy())))
    sess.run(12)                                   # previews after frames 6 and 12
    sess.move_camera(mouse_dx=2.0)                 # one frame of motion: the accumulation restarts
    sess.run(6)                                    # preview after frame 18
    if rank == 0:
        np.savez(out_path, frames=np.array([n for n, _ in seen]), **{"f%d" % i: f for i, (_, f) in enumerate(seen)})
    else:
        assert not seen
    dist.barrier(); dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_progressive_session_two_ranks(tmp_path, pkg, oracle, cornell_scene):
    import torch.multiprocessing as mp
    W, H, P, world = 32, 18, 1024, 2
    out = str(tmp_path / "previews.npz")
    mp.spawn(_worker, args=(world, _free_port(), W, H, P, out), nprocs=world, join=True)
    z = np.load(out)
    assert z["frames"].tolist() == [6, 12, 18]
    # replay on one process: every preview is the per-band oracle state at that frame, driven by the same host camera + events
    cams, orcs = [], []
    for rank, (y0, rows) in enumerate(pkg.tiles.row_bands(H, world)):
        cam = pkg.capi.Camera(W, H); cam.set_pose(*cornell_scene["camera"])
        cams.append(cam); orcs.append(oracle.Renderer(cornell_scene, W, rows, P, tile=(0, y0)))
    for frame in range(1, 19):
        for cam, orc in zip(cams, orcs):
            if frame == 13:
                cam.set_input(mouse_dx=2.0)
            cam.update(0.0); orc.set_camera(cam.buffer); orc.iterate()
        if frame % 6 == 0:
            full = np.concatenate([o.framebuffer() for o in orcs], axis=0)
            assert np.array_equal(z["f%d" % (frame // 6 - 1)].view(np.uint32), full.view(np.uint32))
    spp = [int(z["f%d" % i][..., 3].view(np.uint32).sum()) for i in range(3)]
    assert spp[1] > spp[0] > 0 and spp[2] < spp[1]          # the camera move cleared the accumulation (logic.hlsl:206)
    img = pkg.progressive.to_rgba8(z["f1"])
    assert img.dtype == np.uint8 and img.shape == (H, W, 4) and np.all(img[..., 3] == 255) and img[..., :3].max() <= 186   # tonemapped <= 0.73
    for o in orcs:
        o.close()
