"""GPU tests at BASELINE.json's full sizes (config 3: ~260k triangles, 1920x1080, pool 2^21).

A few iterations are compared bit for bit with the oracle (its ray casts run on the box's host cores); a longer run is
checked through size-independent properties: the extension queue is a permutation of the live slots, counters are
consistent, every completed path lands in exactly one pixel, the per-sample tonemap bound holds, and two independent
renderers agree bit for bit (no schedule dependence).
"""
import numpy as np
import pytest

import oracle_lib as O
import parity_util as PU

pytestmark = pytest.mark.gpu
W, H, P = 1920, 1080, 1 << 21


@pytest.fixture(scope="module")
def big_scene(pkg):
    scene = pkg.scenes.build_scene(pkg.scenes.spheres_mesh(202, 3, seed=1234))
    assert 254000 < scene["num_triangles"] < 266000          # "~260k triangles" +- 2 %
    return scene


def test_fullsize_first_iterations_bitwise(pkg, device, big_scene):
    orc, hip, ocam, hcam, sb = PU.make_pair(pkg, device, big_scene, W, H, P, tile=(0, 0), threads=16)
    for it in range(3):
        PU.step_both(orc, hip, ocam, hcam)
    bad = PU.compare_state(orc, hip, P, P)
    assert not bad, bad[:4]
    assert np.array_equal(orc.counters(), hip.counters())
    assert np.array_equal(orc.framebuffer().view(np.uint32), hip.framebuffer().view(np.uint32))
    assert (hip.stats().flags & 1) == 0
    hip.close(); sb.close(); orc.close()


def test_fullsize_properties_and_determinism(pkg, device, big_scene):
    capi = pkg.capi
    sb = capi.SceneBuffers(device, big_scene)
    rs = []
    for _ in range(2):
        r = capi.Renderer(device, W, H, pool_paths=P, tile=(0, 0))
        r.bind_scene(sb)
        cam = capi.Camera(W, H); cam.set_pose(*big_scene["camera"])
        for _ in range(40):
            cam.update(0.0); r.set_camera(cam.buffer); r.iterate()
        rs.append(r)
    a, b = rs
    fa, fb = a.framebuffer(), b.framebuffer()
    assert np.array_equal(fa.view(np.uint32), fb.view(np.uint32)), "two renderers disagree: schedule dependence"
    assert np.array_equal(a.read_path_state(), b.read_path_state())
    qc = a.counters(); q = a.read_queues(); st = a.stats()
    assert qc[7] == P and qc[0] == 0 and qc[2] == 0 and qc[3] == 0
    assert np.array_equal(np.sort(q[3]), np.arange(P, dtype=np.uint32)), "extension queue must be a permutation of the pool"
    assert int(fa[..., 3].view(np.uint32).sum()) == st.paths_completed
    assert st.paths_generated == P + st.paths_completed and st.segments == 39 * P
    assert np.nanmax(fa[..., :3]) <= 0.5 ** (1 / 2.2) + 1e-6 and not np.isnan(fa).any()
    assert (st.flags & 1) == 0
    a.close(); b.close(); sb.close()


def test_bench_scene_long_run_bitwise(pkg, device, big_scene):
    # the bench scene (259 372 triangles) at a pool the oracle can follow for two completion periods: deep paths, glass, metal,
    # Russian roulette, the reference's immortal NaN paths -- everything the 64-spp render meets, bit for bit
    Wm, Hm, Pm = 256, 144, 1 << 14
    orc, hip, ocam, hcam, sb = PU.make_pair(pkg, device, big_scene, Wm, Hm, Pm, threads=16)
    for it in range(430):
        PU.step_both(orc, hip, ocam, hcam)
        if it % 107 == 106 or it == 429:
            bad = PU.compare_state(orc, hip, Pm, Pm)
            assert not bad, (it, bad[:3])
            assert np.array_equal(orc.framebuffer().view(np.uint32), hip.framebuffer().view(np.uint32)), it
    assert np.array_equal(orc.counters(), hip.counters())
    so, sh = orc.stats(), hip.stats()
    assert so.pathsEnded == sh.paths_completed > Pm and so.segments == sh.segments
    pl = O.state_field(orc.path_state(), Pm, "pathLength")
    assert int(pl.max()) > 150
    hip.close(); sb.close(); orc.close()


def test_config2_cornell_720p_reference_constants_bitwise(pkg, device, cornell_scene):
    # BASELINE config 2 at its real size: Cornell box (34 triangles), 1280x720, the reference's PATHCOUNT = 2^21 with only
    # ITERATIONS * NUM_GROUPS * NUM_THREADS = 2 088 960 live slots (quirk Q1), no tile (the literal uint(1 / pixelSize.x) width, exact at
    # 1280), default camera and lights.  1 spp needs 921 600 paths: frame 0 starts 2 088 960 of them, so the first iterations carry
    # more than one sample per pixel; four iterations are compared bit for bit (state, queues, counters, framebuffer).
    W, H, P, L = 1280, 720, 1 << 21, 2088960
    orc, hip, ocam, hcam, sb = PU.make_pair(pkg, device, cornell_scene, W, H, P, live=L, threads=16)
    for it in range(4):
        PU.step_both(orc, hip, ocam, hcam)
        if it in (0, 3):
            bad = PU.compare_state(orc, hip, P, L)
            assert not bad, "iteration %d: %r" % (it, bad[:3])
            assert np.array_equal(orc.counters(), hip.counters())
    fa, fb = orc.framebuffer(), hip.framebuffer()
    assert np.array_equal(fa.view(np.uint32), fb.view(np.uint32))
    n_ext = int(orc.counters()[7])
    assert np.array_equal(orc.queues()[3][:n_ext], hip.read_queues()[3][:n_ext])
    spp = fb[..., 3].view(np.uint32)
    assert spp.sum() > 0 and hip.counters()[1] >= L          # lastPathCnt advanced by PATHCOUNT on the clear frame and by the regenerated paths since
    hip.close(); sb.close(); orc.close()
