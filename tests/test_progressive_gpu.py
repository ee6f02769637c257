"""f4 on the GPU: the progressive session over the C-ABI renderer (one rank): previews, light edit, resolution switch."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_progressive_session_events(pkg, device, oracle, cornell_scene):
    W, H, P = 48, 27, 4096
    sb = pkg.capi.SceneBuffers(device, cornell_scene)
    r = pkg.capi.Renderer(device, W, H, pool_paths=P); r.bind_scene(sb)
    cam = pkg.capi.Camera(W, H); cam.set_pose(*cornell_scene["camera"])
    orc = oracle.Renderer(cornell_scene, W, H, P)
    seen = []
    sess = pkg.progressive.ProgressiveSession(r, cam, W, H, preview_every=5, on_preview=lambda n, f: seen.append(n))
    ocam = pkg.capi.Camera(W, H); ocam.set_pose(*cornell_scene["camera"])

    def oracle_frames(n):
        for _ in range(n):
            ocam.update(0.0); orc.set_camera(ocam.buffer); orc.iterate()
    frame = sess.run(10)
    oracle_frames(10)
    assert seen == [5, 10] and np.array_equal(frame.view(np.uint32), orc.framebuffer().view(np.uint32))
    # light edit (GUI.cpp:125-130): new emission / position, one light only; both sides restart their accumulation
    lights = cornell_scene["lights"].copy()
    lights[0]["position"] = (2.0, 6.0, 1.0); lights[0]["emission"] = (20.0, 90.0, 20.0); lights[1]["emission"] = 0
    sess.set_lights(sb.lights, lights, 1)
    scene2 = dict(cornell_scene); scene2["lights"] = lights
    orc2 = oracle.Renderer(scene2, W, H, P)
    # the oracle is restarted with the path pool of the running render, as the edit does not touch the pool
    orc2.path_state()[:] = orc.path_state(); orc2.queues()[:] = orc.queues(); orc2.counters()[:] = orc.counters(); orc2.framebuffer()[:] = orc.framebuffer()
    ocam.buffer.lightCount = 1; ocam.reset_accumulation()
    frame = sess.run(5)
    for _ in range(5):
        ocam.update(0.0); orc2.set_camera(ocam.buffer); orc2.iterate()
    assert np.array_equal(frame.view(np.uint32), orc2.framebuffer().view(np.uint32))
    assert frame[..., 1].mean() > frame[..., 0].mean()          # the green light now dominates
    # resolution switch (Renderer.cpp:408-413)
    sess.resize(32, 18)
    frame = sess.run(5)
    assert frame.shape == (18, 32, 4) and int(frame[..., 3].view(np.uint32).max()) >= 1
    assert int(cam.buffer.iterationCounter) == 4                 # restarted at the switch: frames 0..4
    r.close(); sb.close(); orc.close(); orc2.close()
