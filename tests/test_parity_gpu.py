"""GPU parity tests: the HIP pipeline (through the C-ABI, libgmupt.so) against the CPU oracle on the same seeded inputs.

Bar: BIT-EXACT path state (all 21 fields of the reference layout), queue contents, counters and RGBA32F framebuffer.
north_star asks for per-pixel radiance within 1e-4 relative; both sides evaluate the same stated sequence of IEEE
binary32 operations, so the tests assert equality of the bit patterns (tolerance 0 <= 1e-4).
The oracle itself is "parity unpinned" against the DX11 reference (no golden vectors exist; see oracle/gmupt_oracle.h).
"""
import glob
import os

import numpy as np
import pytest

import oracle_lib as O
import parity_util as PU

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REL_TOL = 1e-4  # north_star tolerance; the tests below are stricter (bitwise)


@pytest.fixture(autouse=True, params=["wide", "cast0"])
def shipped_kernel(request, monkeypatch):
    """Every test of this file runs with both shipped fused ray-cast kernels: "wide" (the default: k_cast_w over the 4-wide collapse of the
    tree) and "cast0" (k_cast_f over the binary tree, which also takes whatever the wide kernel does not).  A test that selects a rung
    itself overrides the variable."""
    if request.param != "wide" and any(k in request.node.name for k in ("traversal_rung", "detmath", "shipped_library_has_only", "optin_pruning", "error_convention")):
        pytest.skip("selects its own kernel, or does not cast rays")
    monkeypatch.setenv("GMUPT_TRAVERSAL", request.param)
    return request.param


def _assert_same(orc, hip, P, L, it, check_queues=True):
    bad = PU.compare_state(orc, hip, P, L)
    assert not bad, "iteration %d: path state differs: %r" % (it, bad[:4])
    qa, qb = orc.counters(), hip.counters()
    assert np.array_equal(qa, qb), "iteration %d: counters %r vs %r" % (it, qa.tolist(), qb.tolist())
    fa, fb = orc.framebuffer(), hip.framebuffer()
    assert PU.max_rel_err(fa[..., :3], fb[..., :3]) <= REL_TOL
    assert np.array_equal(fa.view(np.uint32), fb.view(np.uint32)), "iteration %d: framebuffer differs" % it
    if check_queues:
        oq, hq = orc.queues(), hip.read_queues()
        n_ext = int(qa[7])
        assert np.array_equal(oq[3][:n_ext], hq[3][:n_ext]), "extension queue"
        # the shadow queue is a set: its order does not influence any result (no RNG in shadowRayCast)
        assert sorted(oq[4][:qa[6]].tolist()) == sorted(hq[4][:qb[6]].tolist()), "shadow queue"


def test_detmath_bit_identical(device):
    rng = np.random.default_rng(0)
    n = 1 << 18
    cases = [(0, rng.uniform(-400, 400, n), None), (1, rng.uniform(-400, 400, n), None), (0, rng.uniform(-7, 7, n), None),
             (2, np.exp(rng.uniform(-90, 3, n)), None), (3, rng.uniform(-160, 130, n), None),
             (4, rng.uniform(0, 1, n), np.full(n, 1 / 2.2)), (5, rng.uniform(-100, 100, n), None),
             (6, rng.integers(0, 1 << 22, n).astype(np.float32), rng.uniform(0, 1, n))]
    for fn, x, y in cases:
        a, b = O.detmath(fn, x, y), device.detmath(fn, x, y)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), "detmath fn %d differs in %d of %d" % (fn, (a.view(np.uint32) != b.view(np.uint32)).sum(), n)
    sp = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1e-45, 1e38, -1e-10, 3.4e38], np.float32)
    for fn in (0, 1, 3, 5):
        a, b = O.detmath(fn, sp), device.detmath(fn, sp)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), fn


@pytest.mark.parametrize("scene_name,W,H,P,L,iters", [
    ("cornell", 64, 36, 4096, 0, 30),          # config 2 at fixture size: diffuse only, 2 default lights
    ("cornell", 32, 18, 8192, 6144, 12),       # reference quirk Q1: live < pool, 14 in-flight paths per pixel (accumulation order)
    ("soup", 48, 27, 2048, 0, 40),             # UE4 dielectric + metal + glass, spatial-split references
    ("spheres", 48, 27, 2048, 0, 120),         # closed room: long paths, glass spheres
    ("textured", 48, 27, 2048, 0, 40),         # base colour / metallic-roughness / normal-map texture arrays (logic.hlsl:99-124)
])
def test_iteration_parity(pkg, device, cornell_scene, soup_scene, spheres_small_scene, scene_name, W, H, P, L, iters):
    scene = {"cornell": cornell_scene, "soup": soup_scene, "spheres": spheres_small_scene}[scene_name] if scene_name != "textured" \
        else pkg.scenes.build_scene(pkg.scenes.textured_mesh())
    orc, hip, ocam, hcam, sb = PU.make_pair(pkg, device, scene, W, H, P, live=L)
    live = L or P
    for it in range(iters):
        PU.step_both(orc, hip, ocam, hcam)
        if it < 6 or it % 10 == 9 or it == iters - 1:
            _assert_same(orc, hip, P, live, it)
    so, sh = orc.stats(), hip.stats()
    assert so.pathsEnded == sh.paths_completed and so.pathsGenerated == sh.paths_generated and so.segments == sh.segments
    assert (sh.flags & 1) == 0, "traversal stack overflow flag"
    hip.close(); sb.close(); orc.close()


@pytest.mark.parametrize("pad", ["0", "64", "4160"])
def test_path_state_field_stride(pkg, device, monkeypatch, soup_scene, pad):
    # the fields of the path state lie P + GMUPT_STATE_PAD words apart (default 1088: gmupt_renderer_create); nothing depends on the distance --
    # kernels, gmupt_debug_read / write_path_state (the stage-level tests write a frozen state) -- checked at other distances, incl. none
    monkeypatch.setenv("GMUPT_STATE_PAD", pad)
    W, H, P = 48, 27, 2048
    orc, hip, ocam, hcam, sb = PU.make_pair(pkg, device, soup_scene, W, H, P)
    for it in range(12):
        PU.step_both(orc, hip, ocam, hcam)
        _assert_same(orc, hip, P, P, it)
    frozen = orc.path_state().copy()
    hip.write_path_state(frozen)                                      # round trip through the strided device layout
    assert not PU.compare_state(orc, hip, P, P)
    hip.close(); sb.close(); orc.close()


@pytest.mark.parametrize("mode", ["ref", "static", "whilewhile", "ifif1", "top", "coop", "def0", "def1", "pipe0", "cast0", "cast1", "cast2", "cast3", "wide"])
def test_every_traversal_rung_gives_the_same_bits(pkg, soup_scene, monkeypatch, mode):
    # the ladder of ray-cast kernels kept for A/B timing (GMUPT_TRAVERSAL): every rung against the oracle.
    # The rungs are only part of the -DGMUPT_VARIANTS test build of the library (libgmupt_variants.so); the shipped one has cast0 + def0.
    monkeypatch.setenv("GMUPT_TRAVERSAL", mode)
    W, H, P = 48, 27, 4096
    with pkg.capi.use_build("variants"):
        dev = pkg.capi.Device(0)
        orc, hip, ocam, hcam, sb = PU.make_pair(pkg, dev, soup_scene, W, H, P)
        for it in range(14):
            PU.step_both(orc, hip, ocam, hcam)
        _assert_same(orc, hip, P, P, 14)
        hip.close(); sb.close(); orc.close(); dev.close()


def test_shipped_library_has_only_the_shipped_rungs(pkg, device, soup_scene, monkeypatch):
    # the default build refuses a rung it does not contain instead of silently running another kernel
    monkeypatch.setenv("GMUPT_TRAVERSAL", "coop")
    with pytest.raises(pkg.capi.GmuptError, match="not part of this build"):
        pkg.capi.Renderer(device, 16, 16, pool_paths=256)
    monkeypatch.setenv("GMUPT_TRAVERSAL", "def0")
    W, H, P = 48, 27, 4096
    orc, hip, ocam, hcam, sb = PU.make_pair(pkg, device, soup_scene, W, H, P)
    for it in range(14):
        PU.step_both(orc, hip, ocam, hcam)
    _assert_same(orc, hip, P, P, 14)
    assert (hip.stats().flags & pkg.capi.STAT_FUSED_CAST) == 0
    hip.close(); sb.close(); orc.close()


def test_rank_computation_with_many_groups(pkg, spheres_small_scene):
    # k_logic / k_material rank the queue entries in two levels: class counts per block, their totals per group of kScanGroup blocks.
    # A block adds up the totals of the groups before its own with a loop strided by the block size (256): pools beyond 2^22 (config 5:
    # 512 groups) take more than one trip.  The scan1 test build has one block per group, so a pool of 2^17 slots has 512 groups and
    # reaches that path in seconds: state, queues, counters and framebuffer bit for bit against the oracle.
    W, H, P = 160, 90, 1 << 17
    with pkg.capi.use_build("scan1"):
        dev = pkg.capi.Device(0)
        orc, hip, ocam, hcam, sb = PU.make_pair(pkg, dev, spheres_small_scene, W, H, P, threads=16)
        for it in range(8):
            PU.step_both(orc, hip, ocam, hcam)
            if it in (0, 1, 4, 7):
                _assert_same(orc, hip, P, P, it)
        hip.close(); sb.close(); orc.close(); dev.close()


@pytest.mark.parametrize("knobs", [
    {"GMUPT_RAYS_PER_WAVE": "64", "GMUPT_REFILL": "1", "GMUPT_TRI_THRESH": "1", "GMUPT_WAVES_PER_CU": "4"},
    {"GMUPT_RAYS_PER_WAVE": "128", "GMUPT_REFILL": "64", "GMUPT_TRI_THRESH": "64", "GMUPT_WAVES_PER_CU": "16", "GMUPT_TOP_ORDER": "bfs"},
], ids=["eager", "lazy"])
def test_results_do_not_depend_on_scheduling_knobs(pkg, device, spheres_small_scene, monkeypatch, knobs):
    # chunk size, refill threshold, triangle-burst threshold, grid size and the choice of LDS-resident nodes change how the ray casts
    # are scheduled, never what they compute
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    W, H, P = 48, 27, 4096
    orc, hip, ocam, hcam, sb = PU.make_pair(pkg, device, spheres_small_scene, W, H, P)
    for it in range(16):
        PU.step_both(orc, hip, ocam, hcam)
    _assert_same(orc, hip, P, P, 16)
    hip.close(); sb.close(); orc.close()


def test_deep_tree_uses_stack_overflow_path(pkg, device):
    # tree depth ~30: deeper than the LDS part of the traversal stacks (and than the reference's unchecked 16 entries, Q23)
    scene = pkg.scenes.build_scene(pkg.scenes.deep_chain_mesh())
    assert scene["depth"] > 24
    W, H, P = 48, 32, 2048
    orc, hip, ocam, hcam, sb = PU.make_pair(pkg, device, scene, W, H, P)
    for it in range(12):
        PU.step_both(orc, hip, ocam, hcam)
        _assert_same(orc, hip, P, P, it)
    assert orc.stats().maxStack > 24, "rays along the chain must stack more deferred nodes than the LDS part of the stack holds"
    assert (hip.stats().flags & 1) == 0
    hip.close(); sb.close(); orc.close()


def test_edge_cases_odd_pool_single_triangle_and_no_geometry_hit(pkg, device):
    # pool / live counts that are not multiples of the workgroup size, a BVH whose root is a leaf, rays that miss everything (envColor path)
    capi = pkg.capi
    mesh = pkg.scenes.cornell_mesh()
    one = {**mesh, "verts": np.array([[-3, 0, -2], [3, 0, -2], [0, 5, -2]], np.float32), "normals": np.array([[0, 0, 1]] * 3, np.float32),
           "vertex_material": np.zeros(3, np.uint32), "indices": np.array([[0, 1, 2]], np.int32), "name": "one_triangle"}
    scene = pkg.scenes.build_scene(one)
    assert scene["nodes"].shape[0] == 1 and scene["nodes"]["isLeaf"][0] == 1
    W, H, P, L = 40, 24, 1000, 900
    orc, hip, ocam, hcam, sb = PU.make_pair(pkg, device, scene, W, H, P, live=L)
    for it in range(14):
        PU.step_both(orc, hip, ocam, hcam)
        _assert_same(orc, hip, P, L, it)
    fb = hip.framebuffer()
    assert int(fb[..., 3].view(np.uint32).sum()) == hip.stats().paths_completed > 0
    hip.close(); sb.close(); orc.close()
    # zero lights sampled (lightCount = 0): light 0 of the zero-padded table is picked, nothing crashes, still bit-identical
    scene2 = pkg.scenes.build_scene(mesh); scene2["light_count"] = 0
    orc, hip, ocam, hcam, sb = PU.make_pair(pkg, device, scene2, 32, 18, 777)
    for it in range(8):
        PU.step_both(orc, hip, ocam, hcam)
    _assert_same(orc, hip, 777, 777, 7)
    hip.close(); sb.close(); orc.close()


def test_russian_roulette_branch_is_exercised(pkg, device, spheres_small_scene):
    # closed room: paths live until the Russian roulette of logic.hlsl:248-255 (pathLength > 200) -- a rare branch with its own RNG draw
    # that shifts the three draws of createShadowRay; run past it and require that it was really taken
    W, H, P = 40, 24, 2048
    orc, hip, ocam, hcam, sb = PU.make_pair(pkg, device, spheres_small_scene, W, H, P)
    seen_long = 0
    for it in range(330):
        PU.step_both(orc, hip, ocam, hcam)
        if it >= 199 and it % 10 == 9:
            pl = O.state_field(orc.path_state(), P, "pathLength")
            seen_long = max(seen_long, int(pl.max()))
        if it in (150, 205, 215, 260, 329):
            _assert_same(orc, hip, P, P, it)
    assert seen_long > 200, "no path reached the roulette threshold"
    so, sh = orc.stats(), hip.stats()
    assert so.pathsEnded == sh.paths_completed and so.pathsEnded > P, "every initial path must have ended (most of them by roulette)"
    hip.close(); sb.close(); orc.close()


def test_traversal_statistics_match(pkg, device, soup_scene, shipped_kernel):
    # the counting variant of the ray-cast kernels reports the same inner-node / triangle-test totals as the oracle for the
    # extension stage (the visited set does not depend on the traversal order)
    W, H, P = 32, 18, 1024
    orc, hip, ocam, hcam, sb = PU.make_pair(pkg, device, soup_scene, W, H, P, collect_stats=True)
    for it in range(10):
        PU.step_both(orc, hip, ocam, hcam)
    _assert_same(orc, hip, P, P, 9)
    so, sh = orc.stats(), hip.stats()
    if shipped_kernel == "wide":
        # the wide walk reaches the same leaves and makes the same triangle tests through about half as many (4-wide) nodes; a ray that was
        # walked again in the reference's order (an exact tie) is not counted a second time
        assert sh.flags & pkg.capi.STAT_CAST_WIDE and sh.wide_nodes > 0 and sh.wide_box_tests > 0
        assert (so.extRays, so.extLeaves, so.extTris) == (sh.ext_rays, sh.ext_leaves, sh.ext_tris) and 0 < sh.ext_inner < 0.7 * so.extInner
    else:
        assert (so.extRays, so.extInner, so.extLeaves, so.extTris) == (sh.ext_rays, sh.ext_inner, sh.ext_leaves, sh.ext_tris)
    # the any-hit shadow ray is free in how far it walks (deferred triangle tests walk further, skipping boxes entered beyond the
    # light walks less): only the number of rays and -- through _assert_same above -- every inShadow bit must agree
    assert so.shRays == sh.sh_rays and sh.sh_inner > 0 and sh.sh_tris > 0
    # in the drain of every launch finished lanes walk deferred subtrees of the lanes still busy: it happened, and (above) nothing changed
    assert sh.cast_helper_subtrees > 0
    hip.close(); sb.close(); orc.close()


@pytest.mark.parametrize("var", ["GMUPT_SHADOW_PRUNE", "GMUPT_EXTEND_PRUNE"])
def test_optin_pruning_is_neutral_here(pkg, device, soup_scene, monkeypatch, var):
    # the opt-in distance prunings (off by default, see pt_traverse.hip) leave every bit unchanged on this scene
    W, H, P = 48, 27, 4096
    fbs = []
    for prune in ("0", "1"):
        monkeypatch.setenv(var, prune)
        sb = pkg.capi.SceneBuffers(device, soup_scene)
        r = pkg.capi.Renderer(device, W, H, pool_paths=P); r.bind_scene(sb)
        cam = pkg.capi.Camera(W, H); cam.set_pose(*soup_scene["camera"])
        for _ in range(60):
            cam.update(0.0); r.set_camera(cam.buffer); r.iterate()
        fbs.append((r.framebuffer(), r.read_path_state()))
        r.close(); sb.close()
    assert np.array_equal(fbs[0][0].view(np.uint32), fbs[1][0].view(np.uint32)) and np.array_equal(fbs[0][1], fbs[1][1])


def test_tile_and_budget_and_depth_extensions(pkg, device, cornell_scene):
    # multi-GPU tile (global camera, local accumulation), path budget drain, max depth: bitwise vs the oracle with the same parameters
    W, H, P = 48, 24, 2048
    tile = (0, 8); rows = 8
    budget = W * rows * 6
    orc, hip, ocam, hcam, sb = PU.make_pair(pkg, device, cornell_scene, W, rows, P, tile=tile, path_budget=budget, max_depth=6, full=(W, H))
    it = 0
    while it < 400:
        PU.step_both(orc, hip, ocam, hcam)
        it += 1
        if it < 4 or it % 16 == 0:
            _assert_same(orc, hip, P, P, it, check_queues=False)
        if orc.active_paths() == 0:
            break
    assert orc.active_paths() == 0 and hip.stats().active_paths == 0, "budget did not drain"
    _assert_same(orc, hip, P, P, it, check_queues=False)
    fb = hip.framebuffer()
    spp = fb[..., 3].view(np.uint32)
    assert int(spp.sum()) == budget and np.all(spp == 6), "every pixel of the tile receives exactly 6 samples"
    hip.close(); sb.close(); orc.close()


def test_tile_split_is_statistically_the_single_pipeline_image(pkg, device, cornell_scene):
    # an n-GPU image is not bitwise the 1-GPU image (the RNG seeds derive from local slot / queue indices); it must be the
    # same image up to Monte-Carlo noise.  The yardstick is the difference between two single-pipeline renders with different seeds
    # (another pool size): the two-band render may not differ from the reference render by more than that.
    W, H, spp = 64, 36, 256
    sb = pkg.capi.SceneBuffers(device, cornell_scene)

    def render(width, rows, pool, tile):
        r = pkg.capi.Renderer(device, width, rows, pool_paths=pool, tile=tile, path_budget=width * rows * spp)
        r.bind_scene(sb)
        cam = pkg.capi.Camera(W, H); cam.set_pose(*cornell_scene["camera"]); cam.buffer.lightCount = cornell_scene["light_count"]
        r.render_budget(cam, 100000)
        assert r.stats().paths_completed == width * rows * spp
        fb = r.framebuffer(); r.close()
        assert np.all(fb[..., 3].view(np.uint32) == spp)
        return np.minimum(fb[..., :3].astype(np.float64), 4.0)   # clipped like a displayed image: single very bright samples would dominate an RMSE

    single = render(W, H, 4096, None)
    other_seeds = render(W, H, 3072, None)
    bands = np.concatenate([render(W, rows, 4096, (0, y0)) for y0, rows in pkg.tiles.row_bands(H, 2)], axis=0)
    sb.close()
    assert bands.shape == single.shape and not np.array_equal(bands, single)
    rmse = lambda a, b: float(np.sqrt(np.mean((a - b) ** 2)))
    noise = rmse(single, other_seeds)
    print("rmse single vs bands %.5f, single vs other seeds %.5f" % (rmse(single, bands), noise))
    assert noise > 0.0 and rmse(single, bands) < 1.25 * noise, (rmse(single, bands), noise)
    # no bias: the mean radiance of each band agrees within the noise of a band mean (the per-pixel noise / sqrt(pixels), with slack)
    for y0, rows in pkg.tiles.row_bands(H, 2):
        a, b = single[y0:y0 + rows], bands[y0:y0 + rows]
        assert abs(a.mean() - b.mean()) < 6.0 * noise / np.sqrt(a.size), (y0, a.mean(), b.mean(), noise)


def test_render_budget_helper_matches_oracle(pkg, device, soup_scene):
    g = np.load(os.path.join(GOLDEN, "soup2000_32x18_p1024_i16_budget.npz"))
    W, H, P, budget = int(g["width"]), int(g["height"]), int(g["pool"]), int(g["path_budget"])
    sb = pkg.capi.SceneBuffers(device, soup_scene)
    hip = pkg.capi.Renderer(device, W, H, pool_paths=P, path_budget=budget)
    hip.bind_scene(sb)
    cam = pkg.capi.Camera(W, H); cam.set_pose(*soup_scene["camera"])
    iters = hip.render_budget(cam, 10000)
    assert hip.stats().paths_completed == budget and iters <= int(g["iters"]) + 8
    fb = hip.framebuffer()
    assert np.array_equal(fb[..., 3].view(np.uint32), g["framebuffer"][..., 3].view(np.uint32))
    # iterations past the drain point change nothing, so the golden (64 oracle iterations) is reproduced bit for bit
    assert np.array_equal(fb.view(np.uint32), g["framebuffer"].view(np.uint32))
    hip.close(); sb.close()


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "*_i[0-9]*.npz"))), ids=lambda p: os.path.basename(p)[:-4])
def test_golden_fixtures_on_gpu(pkg, device, path):
    g = np.load(path)
    name = os.path.basename(path)
    mesh = pkg.scenes.cornell_mesh() if name.startswith("cornell") else pkg.scenes.textured_mesh() if name.startswith("textured") else \
        pkg.scenes.spheres_mesh(n_spheres=12, subdiv=2, seed=7, floor_quads=4) if name.startswith("spheres12") else pkg.scenes.random_triangles_mesh(2000, seed=1)
    scene = pkg.scenes.build_scene(mesh)
    W, H, P = int(g["width"]), int(g["height"]), int(g["pool"])
    sb = pkg.capi.SceneBuffers(device, scene)
    hip = pkg.capi.Renderer(device, W, H, pool_paths=P, path_budget=int(g["path_budget"]))
    hip.bind_scene(sb)
    cam = pkg.capi.Camera(W, H); cam.set_pose(*scene["camera"])
    for k in range(int(g["iters"])):
        cam.update(0.0)
        assert (cam.buffer.randomSeed[0], cam.buffer.randomSeed[1]) == tuple(g["seeds"][k])
        hip.set_camera(cam.buffer); hip.iterate()
    assert np.array_equal(hip.framebuffer().view(np.uint32), g["framebuffer"].view(np.uint32))
    assert np.array_equal(hip.counters(), g["counters"])
    hip.close(); sb.close()


def test_stage_level_parity_from_frozen_state(pkg, device, spheres_small_scene):
    # freeze an oracle state mid-render, load it into the HIP renderer and run the three stage groups one by one
    W, H, P = 40, 24, 2048
    orc, hip, ocam, hcam, sb = PU.make_pair(pkg, device, spheres_small_scene, W, H, P)
    for _ in range(25):
        ocam.update(); hcam.update(0.0); orc.set_camera(ocam.buffer); orc.iterate()
    hip.write_path_state(orc.path_state()); hip.write_queues(orc.queues()); hip.write_counters(orc.counters())
    hip.write_framebuffer(orc.framebuffer())
    ocam.update(); hcam.update(0.0); orc.set_camera(ocam.buffer); hip.set_camera(hcam.buffer)
    for s in ("logic", "new_path", "material_ue4", "material_glass"):
        orc.stage(s)
    hip.run_stage(pkg.capi.STAGE_SHADE)
    assert not PU.compare_state(orc, hip, P, P), "shade group (logic + newPath + materialUE4 + materialGlass)"
    assert np.array_equal(orc.framebuffer().view(np.uint32), hip.framebuffer().view(np.uint32))
    qa, qb = orc.counters(), hip.counters()
    assert np.array_equal(qa[[0, 1, 2, 3, 4, 5, 6]], qb[[0, 1, 2, 3, 4, 5, 6]])
    oq, hq = orc.queues(), hip.read_queues()
    for q, n in ((0, qa[0]), (1, qa[2]), (2, qa[3]), (3, P)):
        assert np.array_equal(oq[q][:n], hq[q][:n]), "queue %d" % q
    orc.stage("extension"); hip.run_stage(pkg.capi.STAGE_EXTEND)
    bad = PU.compare_state(orc, hip, P, P, fields=["surfacePoint", "baryCoord", "triangle", "isEmitter", "hitDistance"])
    assert not bad, bad
    orc.stage("shadow"); hip.run_stage(pkg.capi.STAGE_SHADOW)
    assert not PU.compare_state(orc, hip, P, P, fields=["inShadow"])
    assert np.array_equal(orc.counters()[:7], hip.counters()[:7])
    hip.close(); sb.close(); orc.close()


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_ray_casts_on_grid_meshes_with_exact_ties(pkg, device, seed):
    # adversarial geometry for the closest-hit rule: triangles on a coarse integer grid (coplanar duplicates, shared edges and vertices,
    # degenerate triangles) and rays that start on grid points and run along grid directions, so that many hits have EXACTLY equal t
    # and the winner is decided by the order of the tests (strict `t < distance`, extensionRayCast.hlsl:64-74) -- every rung of the
    # default path (fused launch via STAGE_RAYCASTS, separate launches via STAGE_EXTEND / STAGE_SHADOW) against the oracle, bit for bit
    rng = np.random.default_rng(seed)
    n_tris = int(rng.integers(20, 400))
    grid = int(rng.choice([3, 5, 9]))
    nv = max(4, n_tris // 2)
    verts = (rng.integers(0, grid, (nv, 3)) * (8.0 / (grid - 1)) - 4.0).astype(np.float32)
    idx = rng.integers(0, nv, (n_tris, 3)).astype(np.int32)
    idx[: n_tris // 5] = idx[n_tris // 5: 2 * (n_tris // 5)][: n_tris // 5]          # exact duplicates of other triangles
    mesh = pkg.scenes.cornell_mesh()
    normals = np.tile(np.array([0.0, 1.0, 0.0], np.float32), (nv, 1))
    mesh.update({"verts": verts, "normals": normals, "indices": idx, "vertex_material": (rng.integers(0, 3, nv)).astype(np.uint32), "name": "grid%d" % seed})
    mesh.pop("uv", None)
    scene = pkg.scenes.build_scene(mesh)
    P = 4096
    orc, hip, ocam, hcam, sb = PU.make_pair(pkg, device, scene, 32, 18, P)
    pts = (rng.integers(0, grid, (P, 3)) * (8.0 / (grid - 1)) - 4.0).astype(np.float32)
    dirs = rng.integers(-2, 3, (P, 3)).astype(np.float32); dirs[(dirs == 0).all(axis=1)] = (1.0, 0.0, 0.0)
    dirs[: P // 2] /= np.linalg.norm(dirs[: P // 2], axis=1, keepdims=True)             # half of them normalised, half with integer components
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
    hits = O.state_field(orc.path_state(), P, "hitDistance").view(np.float32)[:, 0]
    assert (hits < 3e38).sum() > P // 20
    for stages in ((pkg.capi.STAGE_RAYCASTS,), (pkg.capi.STAGE_EXTEND, pkg.capi.STAGE_SHADOW)):
        hip.write_path_state(frozen[0]); hip.write_queues(frozen[1]); hip.write_counters(frozen[2])
        for sname in stages:
            hip.run_stage(sname)
        bad = PU.compare_state(orc, hip, P, P, fields=fields)
        assert not bad, (stages, bad[:3])
    hip.close(); sb.close(); orc.close()


def _grid_mesh(pkg, rng, n_tris, grid, name):
    nv = max(4, n_tris // 2)
    verts = (rng.integers(0, grid, (nv, 3)) * (8.0 / (grid - 1)) - 4.0).astype(np.float32)
    idx = rng.integers(0, nv, (n_tris, 3)).astype(np.int32)
    idx[: n_tris // 5] = idx[n_tris // 5: 2 * (n_tris // 5)][: n_tris // 5]          # exact duplicates of other triangles
    mesh = pkg.scenes.cornell_mesh()
    mesh.update({"verts": verts, "normals": np.tile(np.array([0.0, 1.0, 0.0], np.float32), (nv, 1)), "indices": idx,
                 "vertex_material": (rng.integers(0, 3, nv)).astype(np.uint32), "name": name})
    mesh.pop("uv", None)
    return mesh


@pytest.mark.parametrize("seed", [21, 22])
def test_drain_protocol_on_tie_meshes(pkg, device, monkeypatch, shipped_kernel, seed):
    # The hand-over protocol of the drain, directed: 64-ray chunks on a launch of 8192 rays of either kind put every wave that gets work
    # into the drain after its first refill; a few thousand coplanar / duplicated triangles on a coarse grid make long walks (deep stacks to
    # give away) AND exact ties in t, which is where the merge rules decide pixels -- k_cast_f: a helper's hit counts only if strictly closer
    # than the owner's, among helpers the later donation wins a tie (`dh > dT`); k_cast_w: equal t between lanes sends the ray to the exact
    # walk.  Asserted: subtrees were given away, helpers gave parts of theirs away in turn (two levels), and every extension AND shadow
    # result equals the oracle's bit for bit.
    monkeypatch.setenv("GMUPT_RAYS_PER_WAVE", "64")
    rng = np.random.default_rng(seed)
    scene = pkg.scenes.build_scene(_grid_mesh(pkg, rng, 3000, 9, "drain%d" % seed))
    P = 8192
    orc, hip, ocam, hcam, sb = PU.make_pair(pkg, device, scene, 32, 18, P, collect_stats=True, threads=16)
    pts = (rng.integers(0, 9, (P, 3)) - 4.0).astype(np.float32)
    dirs = rng.choice(np.array([-2.0, -1.0, -0.5, 0.5, 1.0, 2.0], np.float32), (P, 3))
    dirs[::4] = rng.integers(-2, 3, (P // 4 + (P % 4 > 0), 3)).astype(np.float32)[: len(dirs[::4])]     # a quarter with zero components as well
    dirs[(dirs == 0).all(axis=1)] = (1.0, 0.0, 0.0)
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
    hits = O.state_field(orc.path_state(), P, "hitDistance").view(np.float32)[:, 0]
    occluded = O.state_field(orc.path_state(), P, "inShadow")[:, 0]
    assert (hits < 3e38).sum() > P // 4 and 0 < int((occluded != 0).sum()) < P
    hip.write_path_state(frozen[0]); hip.write_queues(frozen[1]); hip.write_counters(frozen[2])
    hip.run_stage(pkg.capi.STAGE_RAYCASTS)
    bad = PU.compare_state(orc, hip, P, P, fields=["surfacePoint", "baryCoord", "triangle", "isEmitter", "hitDistance", "inShadow"])
    assert not bad, bad[:3]
    sh = hip.stats()
    assert sh.flags & pkg.capi.STAT_FUSED_CAST
    assert sh.cast_helper_subtrees > 0, "no subtree was handed to a free lane"
    assert sh.cast_nested_helpers > 0, "no helper gave a part of its subtree away (two-level donation)"
    print("%s seed %d: %d subtrees given, %d of them by helpers, %d rays walked again" % (shipped_kernel, seed, sh.cast_helper_subtrees, sh.cast_nested_helpers, sh.cast_redo_rays))
    hip.close(); sb.close(); orc.close()


def test_camera_reset_resize_and_light_update(pkg, device, cornell_scene):
    W, H, P = 32, 18, 1024
    orc, hip, ocam, hcam, sb = PU.make_pair(pkg, device, cornell_scene, W, H, P)
    for it in range(6):
        PU.step_both(orc, hip, ocam, hcam)
    # accumulation reset (camera moved / shader reload): iterationCounter back to 0 => clear + regenerate (logic.hlsl:206)
    ocam.c.cb.sampleCounter = -1; hcam.reset_accumulation()
    for it in range(5):
        PU.step_both(orc, hip, ocam, hcam)
        _assert_same(orc, hip, P, P, it)
    # light edit through gmupt_buffer_update (GUI.cpp:125-130)
    lights = cornell_scene["lights"].copy(); lights["emission"][1] = (20.0, 60.0, 90.0); lights["position"][1] = (1.0, 6.0, 0.0)
    sb.lights.update(lights)
    scene2 = dict(cornell_scene); scene2["lights"] = lights
    orc2 = O.Renderer(scene2, W, H, P)
    orc2.path_state()[:] = orc.path_state(); orc2.queues()[:] = orc.queues(); orc2.counters()[:] = orc.counters(); orc2.framebuffer()[:] = orc.framebuffer()
    for it in range(6):
        PU.step_both(orc2, hip, ocam, hcam)
    _assert_same(orc2, hip, P, P, 99)
    # resize: new zeroed accumulation target, camera resolution update resets the counter (Renderer.cpp:408-413, Camera.cpp:22)
    hip.resize(24, 12)
    assert hip.framebuffer().shape == (12, 24, 4) and not hip.framebuffer().any()
    hip.close(); sb.close(); orc.close(); orc2.close()


def test_error_convention_on_device(pkg, device, cornell_scene):
    capi = pkg.capi
    r = capi.Renderer(device, 16, 16, pool_paths=256)
    with pytest.raises(capi.GmuptError, match="no scene bound"):
        r.iterate()
    sb = capi.SceneBuffers(device, cornell_scene)
    r.bind_scene(sb)
    with pytest.raises(capi.GmuptError, match="no camera set"):
        r.iterate()
    with pytest.raises(capi.GmuptError):
        capi.Buffer(device, capi.BUFFER_BVH_NODES, np.zeros(47, np.uint8))        # not a multiple of the 48-byte element
    with pytest.raises(capi.GmuptError):
        capi.Device(99)
    r.close(); sb.close()


def test_ray_cast_watchdog_ends_the_launch_and_flags_it(pkg, device, spheres_small_scene, monkeypatch):
    # the persistent ray-cast kernel must end whatever happens: with the iteration limit set absurdly low every wave gives up at once,
    # the launch returns, the statistics say that its results are invalid AND every call that waits for or hands out a frame fails
    # (GMUPT_ERR_CAST_FAULT) until gmupt_reset_stats; with the default limit the flag stays clear and the same calls succeed
    capi = pkg.capi
    sb = capi.SceneBuffers(device, spheres_small_scene)

    def make(cap, budget=0):
        if cap: monkeypatch.setenv("GMUPT_CAST_LOOP_CAP", str(cap))
        else: monkeypatch.delenv("GMUPT_CAST_LOOP_CAP", raising=False)
        r = capi.Renderer(device, 96, 54, pool_paths=1 << 16, path_budget=budget)
        r.bind_scene(sb)
        cam = capi.Camera(96, 54); cam.set_pose(*spheres_small_scene["camera"]); cam.buffer.lightCount = spheres_small_scene["light_count"]
        return r, cam

    r, cam = make(2)
    for _ in range(6):
        cam.update(0.0); r.set_camera(cam.buffer); r.iterate()
    for call in (r.synchronize, r.framebuffer, r.stats):
        with pytest.raises(capi.GmuptError, match="invalid") as e:
            call()
        assert e.value.code == capi.ERR_CAST_FAULT
    assert r.stats(check=False).flags & capi.STAT_CAST_ABORTED
    # (gmupt_copy_framebuffer_to_device makes the same check; it needs a device pointer from torch, which must be the first HIP user of
    #  its process: tests/test_rccl_gpu.py covers it)
    r.reset_stats()                      # the flag is sticky until the statistics are reset
    r.synchronize()
    r.close()
    r, cam = make(2, budget=96 * 54 * 2)
    with pytest.raises(capi.GmuptError, match="invalid") as e:
        r.render_budget(cam, 1000)
    assert e.value.code == capi.ERR_CAST_FAULT
    r.close()
    r, cam = make(0)
    for _ in range(6):
        cam.update(0.0); r.set_camera(cam.buffer); r.iterate()
    r.synchronize(); r.framebuffer()
    assert not (r.stats().flags & (capi.STAT_CAST_ABORTED | capi.STAT_STACK_OVERFLOW))
    r.close()
    sb.close()
