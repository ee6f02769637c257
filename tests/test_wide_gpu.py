"""GPU parity tests of the wide ray cast (GMUPT_TRAVERSAL=wide: k_cast_w over the 4-wide collapse of the SBVH).

The wide walk visits the reference's leaves in another order and skips some of its box tests (pt_traverse_wide.hip explains why the
leaves are the same, also for rays with zero direction components on flat boxes); what the order could change -- exact ties in t
between two triangles -- is noticed, and those rays are walked again in the reference's binary order inside the same launch.  Bar as everywhere:
path state, queues, counters and framebuffer bit for bit against the CPU oracle.
"""
import numpy as np
import pytest

import oracle_lib as O
import parity_util as PU
from test_parity_gpu import _assert_same

pytestmark = pytest.mark.gpu


@pytest.fixture()
def wide(monkeypatch):
    monkeypatch.setenv("GMUPT_TRAVERSAL", "wide")


def test_wide_statistics(pkg, device, wide, soup_scene):
    # leaves reached and triangles tested by the extension rays are the reference's (the visited set does not depend on the order or on
    # which ancestors are tested); the wide nodes visited are about half the binary inner nodes
    W, H, P = 32, 18, 1024
    orc, hip, ocam, hcam, sb = PU.make_pair(pkg, device, soup_scene, W, H, P, collect_stats=True)
    for it in range(10):
        PU.step_both(orc, hip, ocam, hcam)
    _assert_same(orc, hip, P, P, 9)
    so, sh = orc.stats(), hip.stats()
    assert sh.flags & pkg.capi.STAT_CAST_WIDE
    redo = sh.cast_redo_rays
    assert so.extRays == sh.ext_rays and so.shRays == sh.sh_rays
    if redo == 0:
        assert (so.extLeaves, so.extTris) == (sh.ext_leaves, sh.ext_tris)
    assert 0 < sh.ext_inner < 0.7 * so.extInner
    assert sh.wide_box_tests > 0 and sh.cast_helper_subtrees > 0
    hip.close(); sb.close(); orc.close()


@pytest.mark.parametrize("steps", ["6", "8"])
def test_wide_steps_per_iteration(pkg, device, wide, monkeypatch, soup_scene, steps):
    # the kernel has two instantiations (six / eight steps between two looks at the queues; launch_cast_wide picks eight for tables beyond the
    # Infinity Cache, config 5): both forced here on a small scene, with and without the statistics instantiation, bit for bit against the oracle
    monkeypatch.setenv("GMUPT_WIDE_STEPS", steps)
    for collect in (False, True):
        W, H, P = 32, 18, 1024
        orc, hip, ocam, hcam, sb = PU.make_pair(pkg, device, soup_scene, W, H, P, collect_stats=collect)
        for it in range(8):
            PU.step_both(orc, hip, ocam, hcam)
            _assert_same(orc, hip, P, P, it)
        sh = hip.stats()
        assert sh.flags & pkg.capi.STAT_CAST_WIDE
        if collect:
            assert sh.wide_iterations > 0
        hip.close(); sb.close(); orc.close()


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_wide_on_grid_meshes_with_exact_ties(pkg, device, wide, seed):
    # the adversarial meshes of test_ray_casts_on_grid_meshes_with_exact_ties: coplanar duplicates, shared edges, rays along grid
    # directions from grid points: zero direction components with origins exactly on box planes (flat boxes!), and exact ties
    rng = np.random.default_rng(seed)
    n_tris = int(rng.integers(20, 400))
    grid = int(rng.choice([3, 5, 9]))
    nv = max(4, n_tris // 2)
    verts = (rng.integers(0, grid, (nv, 3)) * (8.0 / (grid - 1)) - 4.0).astype(np.float32)
    idx = rng.integers(0, nv, (n_tris, 3)).astype(np.int32)
    idx[: n_tris // 5] = idx[n_tris // 5: 2 * (n_tris // 5)][: n_tris // 5]
    mesh = pkg.scenes.cornell_mesh()
    normals = np.tile(np.array([0.0, 1.0, 0.0], np.float32), (nv, 1))
    mesh.update({"verts": verts, "normals": normals, "indices": idx, "vertex_material": (rng.integers(0, 3, nv)).astype(np.uint32), "name": "grid%d" % seed})
    mesh.pop("uv", None)
    scene = pkg.scenes.build_scene(mesh)
    P = 4096
    orc, hip, ocam, hcam, sb = PU.make_pair(pkg, device, scene, 32, 18, P, collect_stats=True)
    pts = (rng.integers(0, grid, (P, 3)) * (8.0 / (grid - 1)) - 4.0).astype(np.float32)
    dirs = rng.integers(-2, 3, (P, 3)).astype(np.float32); dirs[(dirs == 0).all(axis=1)] = (1.0, 0.0, 0.0)
    # a third of the rays with all components non-zero (they stay in the wide walk and meet ties), normalised or not
    nz = rng.choice(np.array([-2.0, -1.0, 1.0, 2.0], np.float32), (P, 3))
    third = np.arange(P) % 3 == 0
    dirs[third] = nz[third]
    dirs[: P // 2] /= np.linalg.norm(dirs[: P // 2], axis=1, keepdims=True)
    o = (pts - dirs * rng.integers(1, 4, (P, 1)).astype(np.float32)).astype(np.float32)
    st = orc.path_state()
    O.state_field(st, P, "rayOrigin").view(np.float32)[:] = o; O.state_field(st, P, "rayDirection").view(np.float32)[:] = dirs
    O.state_field(st, P, "shadowrayOrigin").view(np.float32)[:] = o; O.state_field(st, P, "shadowrayDirection").view(np.float32)[:] = dirs
    O.state_field(st, P, "lightDistance").view(np.float32)[:, 0] = rng.uniform(0.5, 12.0, P).astype(np.float32)
    orc.queues()[3][:] = np.arange(P, dtype=np.uint32); orc.queues()[4][:] = np.arange(P, dtype=np.uint32)[::-1]
    qc = orc.counters(); qc[:] = 0; qc[6] = P; qc[7] = P
    ocam.update(); hcam.update(0.0); orc.set_camera(ocam.buffer); hip.set_camera(hcam.buffer)
    frozen = (orc.path_state().copy(), orc.queues().copy(), orc.counters().copy())
    orc.stage("extension"); orc.stage("shadow")
    fields = ["surfacePoint", "baryCoord", "triangle", "isEmitter", "hitDistance", "inShadow"]
    hip.write_path_state(frozen[0]); hip.write_queues(frozen[1]); hip.write_counters(frozen[2])
    hip.run_stage(pkg.capi.STAGE_RAYCASTS)
    bad = PU.compare_state(orc, hip, P, P, fields=fields)
    assert not bad, bad[:3]
    sh = hip.stats()
    assert sh.flags & pkg.capi.STAT_CAST_WIDE
    n_inf = int((dirs == 0).any(axis=1).sum())
    assert sh.cast_redo_rays > 0, "exact ties must go to the exact walk"
    # rays with a zero direction component have an infinite 1 / d: a wave walking one takes the general slab test (the ordered one assumes that
    # the sign of 1 / d tells the near plane from the far one, and (plane - o) * inf can be NaN); test_wide_ties_without_zero_components is the other half
    assert n_inf > 0 and 0 < sh.wide_general_iterations <= sh.wide_iterations, (n_inf, sh.wide_general_iterations, sh.wide_iterations)
    print("grid %d: %d of %d rays with a zero component, %d rays walked again (of %d)" % (seed, n_inf, P, sh.cast_redo_rays, 2 * P))
    hip.close(); sb.close(); orc.close()


def test_wide_ties_without_zero_components(pkg, device, wide):
    # two coplanar, overlapping triangles of different materials and a fan of triangles around a shared vertex: rays with non-zero,
    # exactly representable direction components hit them at bitwise equal t -- the wide walk must notice and defer to the exact walk
    mesh = pkg.scenes.cornell_mesh()
    verts = np.array([[-2, 0, -2], [2, 0, -2], [2, 0, 2], [-2, 0, 2],            # square y = 0, two triangles
                      [-2, 0, -2], [2, 0, -2], [2, 0, 2], [-2, 0, 2],            # the same square again (other material)
                      [0, 1, 0], [1, 1, 0], [0, 1, 1], [-1, 1, 0], [0, 1, -1]],  # a fan at y = 1 around (0, 1, 0)
                     np.float32)
    idx = np.array([[0, 1, 2], [0, 2, 3], [4, 5, 6], [4, 6, 7], [8, 9, 10], [8, 10, 11], [8, 11, 12], [8, 12, 9]], np.int32)
    vm = np.array([0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 2], np.uint32)
    mesh.update({"verts": verts, "normals": np.tile(np.array([0.0, 1.0, 0.0], np.float32), (len(verts), 1)), "indices": idx, "vertex_material": vm, "name": "ties"})
    mesh.pop("uv", None)
    scene = pkg.scenes.build_scene(mesh)
    P = 2048
    orc, hip, ocam, hcam, sb = PU.make_pair(pkg, device, scene, 32, 18, P, collect_stats=True)
    rng = np.random.default_rng(11)
    target = np.zeros((P, 3), np.float32)
    target[:, 0] = rng.integers(-7, 8, P) * 0.25; target[:, 2] = rng.integers(-7, 8, P) * 0.25
    target[P // 2:] = (0.0, 1.0, 0.0)                                            # the fan's shared vertex
    target[P // 2:, 0] += rng.integers(-2, 3, P - P // 2) * 0.25
    dirs = rng.choice(np.array([-1.0, -0.5, 0.5, 1.0], np.float32), (P, 3)); dirs[:, 1] = -np.abs(dirs[:, 1])
    o = (target - dirs * rng.integers(1, 5, (P, 1)).astype(np.float32)).astype(np.float32)
    st = orc.path_state()
    O.state_field(st, P, "rayOrigin").view(np.float32)[:] = o; O.state_field(st, P, "rayDirection").view(np.float32)[:] = dirs
    O.state_field(st, P, "shadowrayOrigin").view(np.float32)[:] = o; O.state_field(st, P, "shadowrayDirection").view(np.float32)[:] = dirs
    O.state_field(st, P, "lightDistance").view(np.float32)[:, 0] = rng.uniform(0.5, 12.0, P).astype(np.float32)
    orc.queues()[3][:] = np.arange(P, dtype=np.uint32); orc.queues()[4][:] = np.arange(P, dtype=np.uint32)
    qc = orc.counters(); qc[:] = 0; qc[6] = P; qc[7] = P
    ocam.update(); hcam.update(0.0); orc.set_camera(ocam.buffer); hip.set_camera(hcam.buffer)
    frozen = (orc.path_state().copy(), orc.queues().copy(), orc.counters().copy())
    orc.stage("extension"); orc.stage("shadow")
    hip.write_path_state(frozen[0]); hip.write_queues(frozen[1]); hip.write_counters(frozen[2])
    hip.run_stage(pkg.capi.STAGE_RAYCASTS)
    bad = PU.compare_state(orc, hip, P, P, fields=["surfacePoint", "baryCoord", "triangle", "isEmitter", "hitDistance", "inShadow"])
    assert not bad, bad[:3]
    sh = hip.stats()
    assert sh.cast_redo_rays > P // 8, "the coplanar squares tie on every ray that hits them: %d redo rays" % sh.cast_redo_rays
    assert sh.wide_general_iterations == 0 and sh.wide_iterations > 0, "every 1 / d is finite here: ordered slab tests only (%d of %d)" % (sh.wide_general_iterations, sh.wide_iterations)
    hip.close(); sb.close(); orc.close()


@pytest.mark.parametrize("scene_name", ["soup", "spheres"])
def test_wide_with_tiny_stacks_parks_rays_for_the_exact_walk(pkg, monkeypatch, soup_scene, spheres_small_scene, scene_name):
    # the wides8 test build gives a lane 8 LDS words for both stacks: the inner stack runs full all the time, the wide walk of such a ray is
    # given up (owner or helper lane, extension or shadow ray) and the exact walk -- LDS share, then the bounds-checked global overflow --
    # takes it: results bit for bit, many rays walked again, no error flag
    monkeypatch.setenv("GMUPT_TRAVERSAL", "wide")
    scene = {"soup": soup_scene, "spheres": spheres_small_scene}[scene_name]
    W, H, P = 48, 27, 4096
    with pkg.capi.use_build("wides8"):
        dev = pkg.capi.Device(0)
        orc, hip, ocam, hcam, sb = PU.make_pair(pkg, dev, scene, W, H, P, collect_stats=True)
        for it in range(20):
            PU.step_both(orc, hip, ocam, hcam)
            if it in (0, 1, 5, 19):
                _assert_same(orc, hip, P, P, it)
        sh = hip.stats()
        assert sh.flags & pkg.capi.STAT_CAST_WIDE and not (sh.flags & (pkg.capi.STAT_STACK_OVERFLOW | pkg.capi.STAT_CAST_ABORTED))
        assert sh.cast_redo_rays > 500, "with 8-word stacks many rays must have been parked: %d of %d" % (sh.cast_redo_rays, sh.ext_rays + sh.sh_rays)
        hip.close(); sb.close(); orc.close(); dev.close()


def test_wide_kernel_declines_a_tree_whose_child_boxes_stick_out(pkg, device, wide, cornell_scene):
    # the wide walk's equivalence with the reference's rests on every child box lying inside its parent's box (pt_traverse_wide.hip); a
    # tree that violates it -- here: one child box of the Cornell tree pushed out through its parent's wall -- gets no wide copy and is
    # walked by the binary-tree kernel, which tests every box where the reference tests it: flags say so, results equal the oracle's
    scene = dict(cornell_scene)
    nodes = cornell_scene["nodes"].copy()
    inner = np.nonzero(nodes["isLeaf"] == 0)[0]
    victim = int(nodes["left"][inner[1]])
    nodes["max"][victim][0] = nodes["max"][inner[1]][0] + 0.75      # sticks out of its parent in +x
    scene["nodes"] = nodes
    W, H, P = 48, 27, 2048
    orc, hip, ocam, hcam, sb = PU.make_pair(pkg, device, scene, W, H, P)
    for it in range(12):
        PU.step_both(orc, hip, ocam, hcam)
    _assert_same(orc, hip, P, P, 12)
    st = hip.stats()
    assert st.flags & pkg.capi.STAT_CAST_FETCH and not (st.flags & pkg.capi.STAT_CAST_WIDE), "flags %#x" % st.flags
    assert st.wide_nodes == 0
    hip.close(); sb.close(); orc.close()
    # the well-formed tree of the same scene does get the wide kernel
    orc, hip, ocam, hcam, sb = PU.make_pair(pkg, device, cornell_scene, W, H, P)
    PU.step_both(orc, hip, ocam, hcam)
    st = hip.stats()
    assert st.flags & pkg.capi.STAT_CAST_WIDE and st.wide_nodes > 0 and st.wide_pairs > 0
    hip.close(); sb.close(); orc.close()
