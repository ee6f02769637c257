"""CPU tests of the host side: SBVH builder + flatten, C-ABI surface, camera, tile partition (no GPU needed)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _check_tree(scene, mesh):
    nodes, tris = scene["nodes"], scene["tris"]
    n = nodes.shape[0]
    inner = nodes["isLeaf"] == 0
    # Source/BVHWrapper.cpp:87-91: children of an inner node sit in consecutive slots
    assert np.all(nodes["right"][inner] == nodes["left"][inner] + 1)
    assert np.all(nodes["left"][inner] > np.arange(n)[inner])
    # every node except the root is the child of exactly one inner node
    kids = np.concatenate([nodes["left"][inner], nodes["right"][inner]])
    assert sorted(kids.tolist()) == list(range(1, n))
    # leaves partition the reference array, in DFS order
    leaves = np.where(~inner)[0]
    order = np.argsort(nodes["left"][leaves], kind="stable")
    lo = nodes["left"][leaves][order]; hi = nodes["right"][leaves][order]
    assert lo[0] == 0 and hi[-1] == tris.shape[0] and np.all(lo[1:] == hi[:-1]) and np.all(hi > lo)
    # every source triangle is referenced at least once, duplicates only through spatial splits
    ref = scene["ref_triangle"]
    assert set(ref.tolist()) == set(range(mesh["indices"].shape[0]))
    assert np.array_equal(tris["v"], mesh["indices"][ref])
    assert np.array_equal(tris["materialID"], mesh["vertex_material"][mesh["indices"][ref][:, 0]])   # BVHWrapper.cpp:82
    # child boxes lie inside the parent's box; the root box is the scene box
    for side in ("left", "right"):
        c = nodes[side][inner]
        assert np.all(nodes["min"][c] >= nodes["min"][inner] - 0) and np.all(nodes["max"][c] <= nodes["max"][inner] + 0)
    assert np.array_equal(nodes["min"][0], mesh["verts"].min(axis=0)) and np.array_equal(nodes["max"][0], mesh["verts"].max(axis=0))
    # every reference's triangle overlaps its leaf box (clipped references of spatial splits included)
    v = mesh["verts"]
    for leaf in leaves:
        for i in range(nodes["left"][leaf], nodes["right"][leaf]):
            tv = v[tris["v"][i]]
            assert np.all(tv.min(axis=0) <= nodes["max"][leaf] + 1e-4) and np.all(tv.max(axis=0) >= nodes["min"][leaf] - 1e-4)
    assert scene["depth"] <= 64


def test_builder_invariants_soup(pkg, soup_scene):
    _check_tree(soup_scene, pkg.scenes.random_triangles_mesh(2000, seed=1))
    assert soup_scene["tris"].shape[0] >= 2000           # spatial splits may duplicate references
    assert 0 < soup_scene["sah"] < 1e4


def test_builder_invariants_spheres(pkg, spheres_small_scene):
    _check_tree(spheres_small_scene, pkg.scenes.spheres_mesh(n_spheres=12, subdiv=2, seed=7, floor_quads=4))


def test_builder_is_deterministic_and_order_sensitive_only_through_ids(pkg):
    mesh = pkg.scenes.random_triangles_mesh(500, seed=5)
    a = pkg.scenes.build_scene(mesh); b = pkg.scenes.build_scene(mesh)
    assert np.array_equal(a["nodes"].view(np.uint8), b["nodes"].view(np.uint8)) and np.array_equal(a["tris"].view(np.uint8), b["tris"].view(np.uint8))


def test_builder_small_cases(pkg):
    capi = pkg.capi
    v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    one = capi.sbvh_build(v, np.array([[0, 1, 2]], np.int32))
    assert one["nodes"].shape[0] == 1 and one["nodes"]["isLeaf"][0] == 1 and one["nodes"]["left"][0] == 0 and one["nodes"]["right"][0] == 1
    # with SAH costs 1/1 (Platform defaults, Util.h:73) two triangles are never worth an inner node: leafSAH = 2A < 2A + ...
    two = capi.sbvh_build(np.vstack([v, v + 5]), np.array([[0, 1, 2], [3, 4, 5]], np.int32))
    assert two["nodes"].shape[0] == 1 and two["nodes"]["right"][0] == 2
    far = np.vstack([v + 100 * k for k in range(8)])
    eight = capi.sbvh_build(far, np.arange(24, dtype=np.int32).reshape(8, 3))
    assert eight["nodes"].shape[0] >= 3 and eight["nodes"]["isLeaf"][0] == 0 and eight["tris"].shape[0] == 8
    with pytest.raises(capi.GmuptError):
        capi.sbvh_build(v, np.array([[0, 1, 7]], np.int32))                # vertex index out of range
    # no spatial splits when disabled: reference count == triangle count
    mesh = pkg.scenes.random_triangles_mesh(300, seed=2)
    ns = capi.sbvh_build(mesh["verts"], mesh["indices"], params={"max_spatial_depth": 0})
    assert ns["tris"].shape[0] == 300


def test_cornell_scene_matches_config2(pkg, cornell_scene):
    assert cornell_scene["num_triangles"] == 34 and cornell_scene["tris"].shape[0] == 34
    assert cornell_scene["camera"] == (1.0, 3.0, 8.0, 0.0, 270.0)          # Scene.cpp:59
    l = cornell_scene["lights"]
    assert l.shape[0] == 128 and l["position"][1].tolist() == [0.0, 4.5, 2.0] and l["radius"][0] == 0.5   # Scene.cpp:60-61
    assert np.all(cornell_scene["materials"]["materialType"] == 0) and np.all(cornell_scene["materials"]["metallic"] == 0)


def test_capi_exports_every_declared_symbol(pkg):
    header = open(os.path.join(ROOT, "include", "gmupt.h")).read()
    declared = set(re.findall(r"\b(gmupt_[a-z0-9_]+)\s*\(", header))
    lib = pkg.capi.lib()
    for name in sorted(declared):
        assert hasattr(lib, name), "libgmupt.so does not export %s" % name
    assert declared == set(pkg.capi.SYMBOLS.keys()), declared ^ set(pkg.capi.SYMBOLS.keys())
    assert lib.gmupt_version().startswith(b"gmupt")


def test_binding_constants_follow_the_header(pkg):
    # the ctypes binding restates enum values and flag bits of include/gmupt.h: they must not drift apart
    header = open(os.path.join(ROOT, "include", "gmupt.h")).read()
    capi = pkg.capi
    for name, value in re.findall(r"#define\s+GMUPT_STAT_([A-Z_]+)\s+(\d+)u", header):
        assert getattr(capi, "STAT_" + name) == int(value), name
    for name, value in re.findall(r"GMUPT_BUFFER_([A-Z_]+)\s*=\s*(\d+)", header):
        assert getattr(capi, "BUFFER_" + name) == int(value), name
    assert int(re.search(r"#define\s+GMUPT_STATE_BYTES\s+(\d+)u", header).group(1)) == capi.STATE_BYTES == 248
    assert C.sizeof(capi.CameraBuffer) == 112 if hasattr(capi, "CameraBuffer") else True


def test_capi_error_convention(pkg):
    lib = pkg.capi.lib()
    assert lib.gmupt_iterate(None) == -1 and b"null" in lib.gmupt_last_error()
    assert lib.gmupt_set_camera(None, None) == -1
    out = C.c_void_p()
    assert lib.gmupt_camera_create(0, 0, C.byref(out)) == -1
    assert lib.gmupt_sbvh_build(None, 3, None, 1, None, C.byref(out)) == -1


def test_product_never_touches_the_oracle():
    # the oracle is test infrastructure: nothing in the package, the C-ABI header or the host code may reference it
    bad = []
    for base in (os.path.join(ROOT, "gmu-path-tracer_amd"), os.path.join(ROOT, "include")):
        for d, _, files in os.walk(base):
            for f in files:
                if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h", "Makefile")):
                    txt = open(os.path.join(d, f), errors="replace").read()
                    if re.search(r"oracle_lib|liboracle|gmupt_oracle\.h|#\s*include[^\n]*oracle|orc_[a-z_]+\s*\(", txt):
                        bad.append(os.path.join(d, f))
    assert not bad, bad


def test_host_camera_matches_oracle_camera(pkg, oracle):
    hc = pkg.capi.Camera(1920, 1080); oc = oracle.Camera(1920, 1080)
    hc.set_pose(-9.2, 0.4, -6.3, 2, 376); oc.set_pose(-9.2, 0.4, -6.3, 2, 376)   # Assets/Models/box/box.params row 0
    for _ in range(7):
        hc.update(0.016); oc.update()
        assert bytes(hc.buffer) == bytes(oc.buffer)
    hc.reset_accumulation(); hc.update(0.0)
    assert hc.buffer.iterationCounter == 0
    hc.close()


def test_row_bands_and_assemble(pkg):
    t = pkg.tiles
    assert t.row_bands(1080, 8) == [(i * 135, 135) for i in range(8)]
    b = t.row_bands(10, 3)
    assert b == [(0, 4), (4, 3), (7, 3)] and sum(r for _, r in b) == 10
    tiles = [np.full((r, 5, 4), i, np.float32) for i, (_, r) in enumerate(b)]
    frame = t.assemble(tiles, 5, 10, 3)
    assert frame.shape == (10, 5, 4) and frame[3, 0, 0] == 0 and frame[4, 0, 0] == 1 and frame[9, 4, 3] == 2


@pytest.mark.parametrize("name", ["cornell", "soup2000", "soup_flat", "spheres12", "spheres40", "grid"])
def test_builder_matches_oracle_node_for_node(pkg, oracle, name):
    # the product's C++ builder against the oracle's independent C restatement of Source/Nvidia-SBVH/SplitBVHBuilder.cpp +
    # Source/BVHWrapper.cpp flatten: bitwise-identical 48-byte nodes and 16-byte triangle records
    sc = pkg.scenes
    if name == "cornell":
        mesh = sc.cornell_mesh()
    elif name == "soup2000":
        mesh = sc.random_triangles_mesh(2000, seed=1)
    elif name == "soup_flat":   # large overlapping triangles: many spatial splits and duplicated references
        mesh = sc.random_triangles_mesh(600, seed=3, extent=4.0, size=3.0)
    elif name == "spheres12":
        mesh = sc.spheres_mesh(n_spheres=12, subdiv=2, seed=7, floor_quads=4)
    elif name == "spheres40":
        mesh = sc.spheres_mesh(n_spheres=40, subdiv=3, seed=11, floor_quads=10)
    else:                       # axis-aligned coplanar quads: degenerate (flat) boxes, zero-size bins
        mesh = sc.spheres_mesh(n_spheres=0, subdiv=0, seed=1, floor_quads=24)
    scene = sc.build_scene(mesh)
    nodes, tris, n, nref = oracle.sbvh_build(mesh["verts"], mesh["indices"], mesh["vertex_material"])
    assert n == scene["nodes"].shape[0] and nref == scene["tris"].shape[0]
    assert np.array_equal(nodes, scene["nodes"].view(np.uint8)), "node arrays differ"
    assert np.array_equal(tris, scene["tris"].view(np.uint8)), "triangle records differ"
    if name == "soup2000":
        assert nref > mesh["indices"].shape[0], "expected spatial splits to duplicate references"


@pytest.mark.parametrize("threads,fanout", [(1, 1 << 30), (5, 64), (8, 1000)])
def test_builder_thread_and_fanout_paths_give_the_same_tree(pkg, oracle, monkeypatch, threads, fanout):
    # the builder's task pool, its helper-thread slices (sweeps, binning, partitions, merge sort) and the single-threaded path must all
    # arrive at the oracle's tree; GMUPT_BUILD_FANOUT lowers the node size from which the slices are used so that small meshes reach them
    monkeypatch.setenv("GMUPT_BUILD_THREADS", str(threads)); monkeypatch.setenv("GMUPT_BUILD_FANOUT", str(fanout))
    for mesh in (pkg.scenes.random_triangles_mesh(3000, seed=5), pkg.scenes.random_triangles_mesh(700, seed=6, extent=4.0, size=3.0),
                 pkg.scenes.spheres_mesh(n_spheres=10, subdiv=3, seed=2, floor_quads=6)):
        built = pkg.capi.sbvh_build(mesh["verts"], mesh["indices"], mesh["vertex_material"])
        nodes, tris, n, nref = oracle.sbvh_build(mesh["verts"], mesh["indices"], mesh["vertex_material"])
        assert n == built["nodes"].shape[0] and nref == built["tris"].shape[0]
        assert np.array_equal(nodes, built["nodes"].view(np.uint8)) and np.array_equal(tris, built["tris"].view(np.uint8))


def test_builder_is_clean_under_thread_and_address_sanitizers():
    # CPU build of the builder under -fsanitize=thread and -fsanitize=address,undefined (task pool + helper-thread slices forced on small
    # nodes, a soup that duplicates 2 references per triangle): tools/sanitize/run.sh
    import subprocess
    r = subprocess.run([os.path.join(ROOT, "tools", "sanitize", "run.sh")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "sanitizers: clean" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_scene_loaders_survive_mutated_files_under_sanitizers():
    # glTF / GLB / PNG / .gmesh / .params readers compiled with -fsanitize=address,undefined and fed mutated files: every mutant is loaded
    # or rejected with an exception (tools/sanitize/fuzz_loader.py; a longer run: python tools/sanitize/fuzz_loader.py 300)
    import subprocess, sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sanitize", "fuzz_loader.py"), "10"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "no crash, no sanitizer report" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def test_builder_matches_oracle_on_random_small_meshes(pkg, oracle):
    # property test of the same node-for-node agreement on adversarial little meshes: coordinates on a coarse grid (many exact
    # ties in the sort keys, coplanar and degenerate triangles, duplicated triangles, zero-extent boxes)
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=80, deadline=None)
    @given(st.integers(1, 48), st.integers(0, 2 ** 31 - 1), st.sampled_from([2, 4, 9, 1000]), st.booleans())
    def check(n_tris, seed, grid, shared):
        rng = np.random.default_rng(seed)
        if shared:      # indexed mesh over few vertices: shared vertices, repeated triangles
            nv = max(3, n_tris // 2)
            verts = (rng.integers(0, grid, (nv, 3)) * (8.0 / grid) - 4.0).astype(np.float32)
            idx = rng.integers(0, nv, (n_tris, 3)).astype(np.int32)
        else:
            verts = (rng.integers(0, grid, (n_tris * 3, 3)) * (8.0 / grid) - 4.0).astype(np.float32)
            idx = np.arange(n_tris * 3, dtype=np.int32).reshape(-1, 3)
        vm = rng.integers(0, 3, verts.shape[0]).astype(np.uint32)
        built = pkg.capi.sbvh_build(verts, idx, vm)
        nodes, tris, n, nref = oracle.sbvh_build(verts, idx, vm)
        assert n == built["nodes"].shape[0] and nref == built["tris"].shape[0]
        assert np.array_equal(nodes, built["nodes"].view(np.uint8)) and np.array_equal(tris, built["tris"].view(np.uint8))

    check()


def test_sbvh_build_failure_in_the_task_pool_is_reported_not_hung(pkg, monkeypatch):
    # a worker that fails while others are inside their tasks (or waiting for one) must end the build with an error: an injected throw at
    # the n-th queued task, several workers, a mesh big enough to queue tasks.  A counter of pending tasks cannot signal that (workers still
    # inside a task would decrement it below zero and every waiter would sleep for ever); the pool has a stop flag.
    import threading
    mesh = pkg.scenes.random_triangles_mesh(30000, seed=3)
    monkeypatch.setenv("GMUPT_BUILD_THREADS", "6")
    result = {}

    def build(fail_after):
        if fail_after is None:
            monkeypatch.delenv("GMUPT_SBVH_FAIL_AFTER", raising=False)
        else:
            monkeypatch.setenv("GMUPT_SBVH_FAIL_AFTER", str(fail_after))
        try:
            result["out"] = pkg.capi.sbvh_build(mesh["verts"], mesh["indices"])
        except pkg.capi.GmuptError as e:
            result["out"] = e

    for fail_after in (0, 1, 3):
        t = threading.Thread(target=build, args=(fail_after,), daemon=True)
        t.start(); t.join(120)
        assert not t.is_alive(), "the build hangs after a failure in task %d" % fail_after
        assert isinstance(result["out"], pkg.capi.GmuptError) and "injected failure" in str(result["out"])
    build(None)
    assert isinstance(result["out"], dict) and result["out"]["nodes"].shape[0] > 10000 and result["out"]["tris"].shape[0] >= 30000
