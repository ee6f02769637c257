"""GPU tests of BASELINE.json config 5 at its full size: ~10 M triangles (1 953 icospheres at subdivision 4 + room), 3840x2160,
pool 2^23, max depth 16 -- the one configuration whose BVH (about 1 GB of traversal records) does not fit the 256 MB Infinity Cache.

Bitwise against the oracle for the first iterations (its ray casts run on the box's host cores), then size-independent
properties over a run long enough for the depth limit to end paths, and a reduced-pool run of the same scene that follows the
oracle bit for bit through the depth limit.  The tests also assert WHICH ray-cast kernel ran: k_cast_f with 32-bit record
offsets (the arrays stay below its 2 GiB guard) in the instantiation whose traversal stacks spill to global memory (the tree
is deeper than the LDS part of the stacks).
"""
import os

import numpy as np
import pytest

import oracle_lib as O
import parity_util as PU

pytestmark = pytest.mark.gpu
W, H, P, MAX_DEPTH = 3840, 2160, 1 << 23, 16


@pytest.fixture(scope="module")
def huge_scene(pkg):
    scene = pkg.scenes.build_scene(pkg.scenes.spheres_mesh(1953, 4, seed=1234))
    assert 9_900_000 < scene["num_triangles"] < 10_100_000
    return scene


def _expect_kernel(pkg, stats, scene):
    capi = pkg.capi
    assert stats.flags & capi.STAT_FUSED_CAST, "both ray casts must have run as one launch (flags %#x)" % stats.flags
    if os.environ.get("GMUPT_TRAVERSAL", "wide") == "wide":
        # the default: k_cast_w over the 4-wide collapse (128-byte records, 32-bit offsets: the arrays stay below its 2 GiB guard)
        assert stats.flags & capi.STAT_CAST_WIDE and stats.wide_nodes * 128 < (1 << 31), "k_cast_w must be the kernel that ran (flags %#x)" % stats.flags
    else:
        assert stats.flags & capi.STAT_CAST_FETCH, "k_cast_f must be the kernel that ran (flags %#x)" % stats.flags
        assert scene["depth"] + 2 > 24 and stats.flags & capi.STAT_STACK_SPILL, "depth %d: the spilling instantiation must have run" % scene["depth"]
    assert (stats.flags & capi.STAT_STACK_OVERFLOW) == 0


def test_config5_first_iterations_bitwise(pkg, device, huge_scene):
    orc, hip, ocam, hcam, sb = PU.make_pair(pkg, device, huge_scene, W, H, P, tile=(0, 0), max_depth=MAX_DEPTH, threads=16)
    for it in range(2):
        PU.step_both(orc, hip, ocam, hcam)
    bad = PU.compare_state(orc, hip, P, P)
    assert not bad, bad[:4]
    assert np.array_equal(orc.counters(), hip.counters())
    assert np.array_equal(orc.framebuffer().view(np.uint32), hip.framebuffer().view(np.uint32))
    n_ext = int(orc.counters()[7])
    assert np.array_equal(orc.queues()[3][:n_ext], hip.read_queues()[3][:n_ext])
    _expect_kernel(pkg, hip.stats(), huge_scene)
    hip.close(); sb.close(); orc.close()


def test_config5_properties_through_the_depth_limit(pkg, device, huge_scene):
    capi = pkg.capi
    sb = capi.SceneBuffers(device, huge_scene)
    iters = 24
    rs = []
    for _ in range(2):
        r = capi.Renderer(device, W, H, pool_paths=P, tile=(0, 0), max_depth=MAX_DEPTH)
        r.bind_scene(sb)
        cam = capi.Camera(W, H); cam.set_pose(*huge_scene["camera"]); cam.buffer.lightCount = huge_scene["light_count"]
        for _ in range(iters):
            cam.update(0.0); r.set_camera(cam.buffer); r.iterate()
        rs.append(r)
    a, b = rs
    fa, fb = a.framebuffer(), b.framebuffer()
    assert np.array_equal(fa.view(np.uint32), fb.view(np.uint32)), "two renderers disagree: schedule dependence"
    qa, qb = a.read_queues(), b.read_queues()
    assert np.array_equal(qa[3], qb[3])
    del qb
    qc = a.counters(); st = a.stats()
    assert qc[7] == P and qc[0] == 0 and qc[2] == 0 and qc[3] == 0
    assert np.array_equal(np.sort(qa[3]), np.arange(P, dtype=np.uint32)), "extension queue must be a permutation of the pool"
    assert int(fa[..., 3].view(np.uint32).sum()) == st.paths_completed           # every completed path landed in exactly one pixel
    assert st.paths_generated == P + st.paths_completed and st.segments == (iters - 1) * P
    assert np.nanmax(fa[..., :3]) <= 0.5 ** (1 / 2.2) + 1e-6 and not np.isnan(fa).any()
    # the depth limit is what ends most paths here: nothing older than MAX_DEPTH segments is in flight
    state = a.read_path_state()
    assert int(O.state_field(state, P, "pathLength").max()) <= MAX_DEPTH
    assert st.paths_completed > P // 2, "after %d iterations with max depth %d the first generation must have ended" % (iters, MAX_DEPTH)
    _expect_kernel(pkg, st, huge_scene)
    a.close(); b.close(); sb.close()


def test_config5_scene_reduced_pool_bitwise_through_the_depth_limit(pkg, device, huge_scene):
    # the same 10 M-triangle tree and depth limit at a pool the oracle can follow for two generations of paths
    Wm, Hm, Pm = 480, 270, 1 << 16
    orc, hip, ocam, hcam, sb = PU.make_pair(pkg, device, huge_scene, Wm, Hm, Pm, max_depth=MAX_DEPTH, threads=16)
    for it in range(2 * MAX_DEPTH + 4):
        PU.step_both(orc, hip, ocam, hcam)
        if it in (0, MAX_DEPTH - 1, MAX_DEPTH, MAX_DEPTH + 1, 2 * MAX_DEPTH + 3):
            bad = PU.compare_state(orc, hip, Pm, Pm)
            assert not bad, (it, bad[:3])
            assert np.array_equal(orc.framebuffer().view(np.uint32), hip.framebuffer().view(np.uint32)), it
            assert np.array_equal(orc.counters(), hip.counters()), it
    so, sh = orc.stats(), hip.stats()
    assert so.pathsEnded == sh.paths_completed > Pm and so.segments == sh.segments
    _expect_kernel(pkg, sh, huge_scene)
    hip.close(); sb.close(); orc.close()
