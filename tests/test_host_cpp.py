"""Tests of the C++ host classes (Renderer / Scene / BVHWrapper / Camera mirror of the reference API) through the headless
driver gmu-path-tracer_amd/host/gmupt_render."""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "gmu-path-tracer_amd", "host")
EXE = os.path.join(HOST, "gmupt_render")


@pytest.fixture(scope="module")
def exe(pkg):
    pkg.capi.lib()
    subprocess.run(["make", "-C", HOST, "-s"], check=True)
    return EXE


def test_build_only_matches_python_builder(exe, pkg, cornell_scene, tmp_path):
    out = json.loads(subprocess.run([exe, "--build-only", "--scene", "cornell"], check=True, capture_output=True, text=True).stdout)
    assert out["triangles"] == 34 and out["nodes"] == cornell_scene["nodes"].shape[0] and out["references"] == 34
    assert abs(out["sah"] - cornell_scene["sah"]) < 1e-4
    mesh = pkg.scenes.random_triangles_mesh(1500, seed=9)
    path = str(tmp_path / "soup.gmesh")
    pkg.scenes.save_gmesh(mesh, path, str(tmp_path / "soup.params"))
    out = json.loads(subprocess.run([exe, "--build-only", "--scene", path], check=True, capture_output=True, text=True).stdout)
    ref = pkg.scenes.build_scene(mesh)
    assert out["nodes"] == ref["nodes"].shape[0] and out["references"] == ref["tris"].shape[0] and abs(out["sah"] - ref["sah"]) < 1e-3


def _loaded(exe, pkg, path, tmp_path, name="loaded.gmesh"):
    """What the C++ scene loader makes of a file: (--build-only summary, mesh dict)."""
    dump = str(tmp_path / name)
    out = json.loads(subprocess.run([exe, "--build-only", "--scene", path, "--dump-mesh", dump], check=True, capture_output=True, text=True).stdout)
    return out, pkg.scenes.load_gmesh(dump)


def _expected_smooth_normals(verts, indices, vertex_material):
    """aiProcess_GenSmoothNormals as documented (postprocess.h:169-183): per mesh (= material), the normalised sum of the unit normals of all
    faces with a corner at the same position.  Returns one normal per face corner (n_tris, 3, 3), float32 arithmetic like the loader."""
    tri = verts[indices].astype(np.float32)
    fn = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]).astype(np.float32)
    ln = np.sqrt((fn * fn).sum(1, dtype=np.float32)).astype(np.float32)
    fn = np.where(ln[:, None] > 0, fn / np.where(ln[:, None] > 0, ln[:, None], 1), fn).astype(np.float32)
    sums = {}
    mats = vertex_material[indices[:, 0]]
    for t in range(indices.shape[0]):
        for k in range(3):
            key = (int(mats[t]),) + tuple((tri[t, k] + np.float32(0)).tolist())
            sums[key] = sums.get(key, np.zeros(3, np.float32)) + fn[t]
    out = np.zeros((indices.shape[0], 3, 3), np.float32)
    for t in range(indices.shape[0]):
        for k in range(3):
            v = sums[(int(mats[t]),) + tuple((tri[t, k] + np.float32(0)).tolist())]
            l = np.sqrt((v * v).sum(dtype=np.float32))
            out[t, k] = v / l if l > 0 else v
    return out


def test_gltf_loader_matches_written_mesh(exe, pkg, tmp_path):
    # f1: Scene(device, "x.gltf") path -- geometry, material factors, BLEND -> glass, flipped UVs, .params sibling
    mesh = pkg.scenes.random_triangles_mesh(1500, seed=4)      # three materials, one of them glass
    path = str(tmp_path / "scene.gltf")
    written = pkg.scenes.save_gltf(mesh, path)
    out, loaded = _loaded(exe, pkg, path, tmp_path)
    ref = pkg.scenes.build_scene(written)
    assert out["triangles"] == written["indices"].shape[0] == loaded["indices"].shape[0]
    # same triangles in the same order: corner positions, authored normals, flipped-back UVs and materials, bit for bit
    for key in ("verts", "normals", "uv"):
        assert np.array_equal(loaded[key][loaded["indices"]], written[key][written["indices"]].astype(np.float32)), key
    assert np.array_equal(loaded["vertex_material"][loaded["indices"]], written["vertex_material"][written["indices"]])
    # aiProcess_JoinIdenticalVertices: no two vertices of a mesh carry the same data, and they are numbered in the order of their first use
    rec = np.concatenate([loaded["verts"], loaded["normals"], loaded["uv"], loaded["vertex_material"][:, None].astype(np.float32)], axis=1)
    assert np.unique(rec, axis=0).shape[0] == rec.shape[0]
    first_use = loaded["indices"].reshape(-1)
    _, where = np.unique(first_use, return_index=True)
    assert np.array_equal(first_use[np.sort(where)], np.arange(rec.shape[0]))
    assert out["materials"] == mesh["materials"].shape[0] and out["glass_materials"] == int((mesh["materials"]["materialType"] == 1).sum()) > 0
    assert out["nodes"] == ref["nodes"].shape[0] and out["references"] == ref["tris"].shape[0] and abs(out["sah"] - ref["sah"]) < 1e-3
    # node transform (translation, rotation about y, uniform scale) is baked into the vertices: aiProcess_PreTransformVertices;
    # the file carries no normals: aiProcess_GenSmoothNormals
    ang = np.deg2rad(30.0)
    tr = {"translation": [1.0, 2.0, -3.0], "rotation": [0.0, float(np.sin(ang / 2)), 0.0, float(np.cos(ang / 2))], "scale": [2.0, 2.0, 2.0]}
    path2 = str(tmp_path / "moved.gltf")
    w2 = pkg.scenes.save_gltf(pkg.scenes.cornell_mesh(), path2, node_transform=tr, with_normals=False)
    out2, l2 = _loaded(exe, pkg, path2, tmp_path, "moved.gmesh")
    Rm = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
    moved = (w2["verts"].astype(np.float64) * 2.0) @ Rm.T + np.array([1.0, 2.0, -3.0])
    assert np.allclose(out2["bbox"], np.concatenate([moved.min(0), moved.max(0)]), atol=1e-4)
    assert out2["triangles"] == 34
    assert np.allclose(l2["verts"][l2["indices"]], moved[w2["indices"]], atol=1e-4)
    want = _expected_smooth_normals(l2["verts"], l2["indices"], l2["vertex_material"])
    assert np.allclose(l2["normals"][l2["indices"]], want, atol=2e-6)
    assert np.allclose(np.linalg.norm(l2["normals"], axis=1), 1.0, atol=1e-5)


def test_gltf_loader_smooths_across_seams_and_joins_duplicates(exe, pkg, tmp_path):
    # two quads folded along a shared edge whose vertices are stored TWICE with different uv (a texture seam), no normals in the file:
    # the generated normals on the seam are the same on both sides (same position => smoothed together, postprocess.h:169-183) and the
    # seam vertices stay separate (their uv differ) -- while corners that carry identical data are merged even though the file lists
    # them under different indices (aiProcess_JoinIdenticalVertices, postprocess.h:84-94)
    base = pkg.scenes.cornell_mesh()
    v = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0],         # quad A in z = 0
                  [1, 0, 0], [1, 0, -1], [1, 1, -1], [1, 1, 0],       # quad B in x = 1; its first and last vertex repeat A's edge (1,0,0)-(1,1,0)
                  [0, 0, 0], [1, 1, 0]], np.float32)                   # two exact duplicates of A's vertices 0 and 2 (same uv below)
    uv = np.array([[0, 0], [1, 0], [1, 1], [0, 1], [0, 0], [1, 0], [1, 1], [0, 1], [0, 0], [1, 1]], np.float32)
    idx = np.array([[0, 1, 2], [8, 9, 3], [4, 5, 6], [4, 6, 7]], np.int32)   # triangle 1 uses the duplicates
    mesh = dict(base); mesh.update({"verts": v, "normals": np.zeros_like(v), "uv": uv, "indices": idx, "vertex_material": np.zeros(10, np.uint32), "name": "seam"})
    path = str(tmp_path / "seam.gltf")
    pkg.scenes.save_gltf(mesh, path, with_normals=False)
    out, l = _loaded(exe, pkg, path, tmp_path, "seam.gmesh")
    assert out["triangles"] == 4 and l["verts"].shape[0] == 8           # 10 file vertices - 2 exact duplicates
    n_corner = l["normals"][l["indices"]]; p_corner = l["verts"][l["indices"]]
    on_seam = (p_corner[..., 0] == 1) & (p_corner[..., 2] == 0)
    # every FACE at a position counts once with its unit normal: at (1,0,0) one z-facing and two x-facing triangles meet, at (1,1,0) two and one
    low = on_seam & (p_corner[..., 1] == 0); high = on_seam & (p_corner[..., 1] == 1)
    assert np.allclose(n_corner[low], np.array([2, 0, 1]) / np.sqrt(5), atol=1e-6) and np.allclose(n_corner[high], np.array([1, 0, 2]) / np.sqrt(5), atol=1e-6)
    assert low.sum() == 3 and high.sum() == 3                           # both sides of the seam carry the same normal
    assert np.allclose(n_corner[(p_corner[..., 0] == 0)], [0, 0, 1]) and np.allclose(n_corner[(p_corner[..., 2] == -1)], [1, 0, 0])
    want = _expected_smooth_normals(l["verts"], l["indices"], l["vertex_material"])
    assert np.allclose(n_corner, want, atol=2e-6)


def test_scene_params_registry(exe, pkg, tmp_path):
    # SceneParams::instance / loadScenes / getSceneIndex (Source/Scene.cpp:22-80): every glTF below the models directory, its .params or the defaults
    root = tmp_path / "Models"
    (root / "box").mkdir(parents=True); (root / "deep" / "er").mkdir(parents=True)
    a = pkg.scenes.save_gltf(pkg.scenes.cornell_mesh(), str(root / "box" / "box.gltf"))
    pkg.scenes.save_gltf(pkg.scenes.random_triangles_mesh(50, seed=2), str(root / "deep" / "er" / "soup.gltf"))
    os.remove(str(root / "deep" / "er" / "soup.params"))                                    # no .params: defaults of Scene.cpp:59-61
    (root / "box" / "notes.txt").write_text("not a scene")
    listed = json.loads(subprocess.run([exe, "--models-root", str(root), "--list-scenes"], check=True, capture_output=True, text=True).stdout)
    assert [s["name"] for s in listed] == ["box/box.gltf", "deep/er/soup.gltf"] and [s["index"] for s in listed] == [0, 1]
    assert np.allclose(listed[0]["camera"], a["camera"]) and listed[0]["lights"] == a["light_count"]
    assert listed[1]["camera"] == [1, 3, 8, 0, 270] and listed[1]["lights"] == 2
    ok = subprocess.run([exe, "--models-root", str(root), "--build-only", "--scene", str(root / "box" / "box.gltf")], capture_output=True, text=True)
    assert ok.returncode == 0
    # a scene below the models directory that the registry does not list (only glTF files are scenes) is the reference's runtime_error (Scene.cpp:79)
    pkg.scenes.save_gmesh(pkg.scenes.cornell_mesh(), str(root / "box" / "dump.gmesh"))
    r = subprocess.run([exe, "--models-root", str(root), "--build-only", "--scene", str(root / "box" / "dump.gmesh")], capture_output=True, text=True)
    assert r.returncode != 0 and "Non existing scene box/dump.gmesh" in r.stderr
    # ... while the same file outside the models directory loads with its sibling .params / the defaults (an extension of this build)
    pkg.scenes.save_gmesh(pkg.scenes.cornell_mesh(), str(tmp_path / "dump.gmesh"))
    assert subprocess.run([exe, "--models-root", str(root), "--build-only", "--scene", str(tmp_path / "dump.gmesh")], capture_output=True, text=True).returncode == 0


def _textured_gltf(pkg, tmp_path):
    path = str(tmp_path / "tex.gltf")
    # material 0's base colour image is written 2x larger: three distinct-size rule -> the loader resizes layers to the median size
    written = pkg.scenes.save_gltf(pkg.scenes.textured_mesh(), path, texture_scale={(0, 0): 2, (0, 2): 2, (2, 2): 2})
    arrays = [pkg.capi.texture_array_from_png(f) if f else None for f in written["texture_files"]]
    return path, written, arrays


def test_gltf_textures_are_decoded_indexed_and_resized(exe, pkg, tmp_path):
    # f3: images of baseColorTexture / metallicRoughnessTexture / normalTexture -> layers in material order, common size by the
    # median rule (Scene.cpp:209-244), resized where they differ (:268-285)
    path, written, arrays = _textured_gltf(pkg, tmp_path)
    out = json.loads(subprocess.run([exe, "--build-only", "--scene", path], check=True, capture_output=True, text=True).stdout)
    assert out["texture_indices"] == [[int(v) for v in m["textureIndices"]] for m in written["materials"]]
    assert out["texture_indices"][0] == [0, 0, 0] and out["texture_indices"][1] == [1, -1, -1] and out["texture_indices"][2] == [-1, 1, 1]
    for t, arr in enumerate(arrays):
        info = out["textures"][t]
        assert info["layers"] == arr.shape[0] and info["size"] == arr.shape[1]
        chk = 0
        for v in arr.reshape(-1).tolist():
            chk = (chk * 31 + v) & 0xFFFFFFFFFFFFFFFF
        assert info["checksum"] == chk
    assert [a.shape[1] for a in arrays] == [32, 16, 32]     # {16, 32} -> 32; all 16 -> 16; both normal maps doubled -> 32


def test_glb_container_loads_like_the_gltf(exe, pkg, tmp_path):
    # the same scene as .gltf (+ .bin + .png files) and as one binary .glb (BIN chunk, images as bufferViews): identical build and textures
    path, written, arrays = _textured_gltf(pkg, tmp_path)
    glb = str(tmp_path / "packed.glb")
    pkg.scenes.gltf_to_glb(path, glb)
    a = json.loads(subprocess.run([exe, "--build-only", "--scene", path], check=True, capture_output=True, text=True).stdout)
    b = json.loads(subprocess.run([exe, "--build-only", "--scene", glb], check=True, capture_output=True, text=True).stdout)
    for k in ("triangles", "vertices", "materials", "glass_materials", "nodes", "references", "sah", "bbox", "textures", "texture_indices"):
        assert a[k] == b[k], k
    assert b["textures"][0]["layers"] == 2
    bad = open(glb, "rb").read()
    (tmp_path / "bad.glb").write_bytes(b"glTX" + bad[4:])
    r = subprocess.run([exe, "--build-only", "--scene", str(tmp_path / "bad.glb")], capture_output=True, text=True)
    assert r.returncode != 0 and "GLB" in r.stderr


def test_reference_params_files_parse(exe, tmp_path):
    # the values of the reference's own .params data files (tests/golden/reference_params.json), written back as CSV and parsed by
    # SceneParams::load (Source/Scene.cpp:34-55)
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_params.json")))["scenes"]
    def run(name):
        path = str(tmp_path / (name + ".params"))
        with open(path, "w") as f:
            rows = [ref[name]["camera_xyz_pitch_yaw"]] + ref[name]["lights_xyz_falloff_rgb_radius"]
            f.write("\n".join(", ".join("%g" % v for v in row) for row in rows))       # no trailing newline, like the originals
        return json.loads(subprocess.run([exe, "--params", path], check=True, capture_output=True, text=True).stdout)
    box = run("box_box")
    assert np.allclose(box["camera"], [-9.2, 0.4, -6.3, 2, 376]) and len(box["lights"]) == 1
    assert np.allclose(box["lights"][0], [0, 4.5, 2.0, 100, 80, 80, 40, 0.5])
    bg = run("bunny_glass_scene")
    assert np.allclose(bg["camera"], [1.0, 3.0, 8.0, 0, 270]) and len(bg["lights"]) == 2 and np.allclose(bg["lights"][0][:3], [13.0, 4.5, 4.5])
    for name in ref:
        out = run(name)
        assert np.allclose(out["camera"], ref[name]["camera_xyz_pitch_yaw"]) and np.allclose(out["lights"], ref[name]["lights_xyz_falloff_rgb_radius"])
    # a missing file gives the defaults of Source/Scene.cpp:59-61
    dflt = json.loads(subprocess.run([exe, "--params", "/nonexistent.params"], check=True, capture_output=True, text=True).stdout)
    assert dflt["camera"] == [1, 3, 8, 0, 270] and len(dflt["lights"]) == 2


def test_missing_scene_is_a_runtime_error(exe):
    r = subprocess.run([exe, "--build-only", "--scene", "/nonexistent/x.gmesh"], capture_output=True, text=True)
    assert r.returncode != 0 and "Non existing scene" in r.stderr            # wording of Source/Scene.cpp:79, exit path of main.cpp:19-23


def _oracle_frames(pkg, oracle, loaded, like, W, H, P, frames, textures=None):
    """The oracle on what the C++ loader produced (mesh dump) with the camera / lights of the .params file."""
    mesh = dict(loaded)
    for k in ("lights", "light_count", "camera", "name"):
        mesh[k] = like[k]
    scene = pkg.scenes.build_scene(mesh)
    if textures:
        for key, arr in zip(("tex_diffuse", "tex_metallic_roughness", "tex_normal"), textures):
            scene[key] = arr
    orc = oracle.Renderer(scene, W, H, P, threads=8)
    cam = oracle.Camera(W, H); cam.set_pose(*scene["camera"])
    for _ in range(frames):
        cam.update(); orc.set_camera(cam.buffer); orc.iterate()
    fb = orc.framebuffer().copy()
    orc.close()
    return fb, scene


@pytest.mark.gpu
def test_cpp_renderer_equals_oracle_on_the_loaded_scene(exe, pkg, oracle, device, tmp_path):
    # Renderer::update()/draw() frames through the C++ classes (scene loaded from glTF + .params by the C++ loader, HIP kernels) against the
    # ORACLE run on the loader's own output (gmupt_render --dump-mesh), bit for bit -- and against the same frames driven through the C-ABI
    mesh = pkg.scenes.save_gltf(pkg.scenes.random_triangles_mesh(1500, seed=4), str(tmp_path / "s12.gltf"))
    path = str(tmp_path / "s12.gltf")
    W, H, P, frames = 48, 27, 4096, 20
    dump = str(tmp_path / "fb.f32")
    subprocess.run([exe, "--scene", path, "--size", "%dx%d" % (W, H), "--frames", str(frames), "--pool", str(P), "--live", str(P),
                    "--dump", dump, "--capture", "--pfm", str(tmp_path / "fb.pfm")], check=True, cwd=str(tmp_path))
    fb_cpp = np.fromfile(dump, dtype=np.float32).reshape(H, W, 4)
    _, loaded = _loaded(exe, pkg, path, tmp_path)
    fb_orc, scene = _oracle_frames(pkg, oracle, loaded, mesh, W, H, P, frames)
    assert np.array_equal(fb_cpp.view(np.uint32), fb_orc.view(np.uint32)), "C++ Renderer (HIP) differs from the oracle on the loader's scene"
    assert int(fb_cpp[..., 3].view(np.uint32).sum()) > 0
    sb = pkg.capi.SceneBuffers(device, scene)
    r = pkg.capi.Renderer(device, W, H, pool_paths=P)
    r.bind_scene(sb)
    cam = pkg.capi.Camera(W, H); cam.set_pose(*scene["camera"])
    for _ in range(frames):
        cam.update(0.0); r.set_camera(cam.buffer); r.iterate()
    assert np.array_equal(fb_cpp.view(np.uint32), r.framebuffer().view(np.uint32))
    raw = (tmp_path / "fb.pfm").read_bytes()                                  # f2: raw float export next to the 8-bit capture
    head = b"PF\n%d %d\n-1.0\n" % (W, H)
    assert raw.startswith(head) and np.array_equal(np.frombuffer(raw[len(head):], "<f4").reshape(H, W, 3)[::-1], fb_cpp[..., :3])
    png = tmp_path / "Captures" / "potato0.png"                               # Renderer.cpp:404 naming
    assert png.exists() and png.read_bytes()[:8] == b"\x89PNG\r\n\x1a\n"
    from PIL import Image
    img = np.asarray(Image.open(str(png)))
    assert img.shape == (H, W, 4) and np.array_equal(img[..., :3], (fb_cpp[..., :3] * 255).astype(np.uint8)) and np.all(img[..., 3] == 255)
    r.close(); sb.close()


@pytest.mark.gpu
def test_cpp_renderer_generated_normals_scene_equals_oracle(exe, pkg, oracle, tmp_path):
    # a glTF without normals under a node transform: the loader generates the smooth normals and joins the vertices; the frames of the C++
    # Renderer equal the oracle's on that loaded scene
    ang = np.deg2rad(20.0)
    tr = {"translation": [0.5, 0.0, -1.0], "rotation": [0.0, float(np.sin(ang / 2)), 0.0, float(np.cos(ang / 2))], "scale": [1.0, 1.0, 1.0]}
    src = pkg.scenes.spheres_mesh(n_spheres=6, subdiv=2, seed=3, floor_quads=4)
    path = str(tmp_path / "gen.gltf")
    written = pkg.scenes.save_gltf(src, path, node_transform=tr, with_normals=False)
    W, H, P, frames = 48, 27, 4096, 24
    dump = str(tmp_path / "gen.f32")
    subprocess.run([exe, "--scene", path, "--size", "%dx%d" % (W, H), "--frames", str(frames), "--pool", str(P), "--live", str(P), "--dump", dump], check=True, cwd=str(tmp_path))
    fb_cpp = np.fromfile(dump, dtype=np.float32).reshape(H, W, 4)
    _, loaded = _loaded(exe, pkg, path, tmp_path, "gen.gmesh")
    assert loaded["verts"].shape[0] <= written["verts"].shape[0]
    fb_orc, _ = _oracle_frames(pkg, oracle, loaded, written, W, H, P, frames)
    assert np.array_equal(fb_cpp.view(np.uint32), fb_orc.view(np.uint32))
    assert int(fb_cpp[..., 3].view(np.uint32).sum()) > 0


@pytest.mark.gpu
def test_cpp_renderer_with_gltf_textures_equals_oracle_and_capi_render(exe, pkg, oracle, device, tmp_path):
    # the three Texture2DArrays built by Scene::loadTextures from the glTF's PNG files drive the same pixels as the arrays
    # built from the same files through the C-ABI helpers
    path, written, arrays = _textured_gltf(pkg, tmp_path)
    W, H, P, frames = 40, 30, 2048, 24
    dump = str(tmp_path / "fb_tex.f32")
    subprocess.run([exe, "--scene", path, "--size", "%dx%d" % (W, H), "--frames", str(frames), "--pool", str(P), "--live", str(P), "--dump", dump], check=True, cwd=str(tmp_path))
    fb_cpp = np.fromfile(dump, dtype=np.float32).reshape(H, W, 4)
    scene = pkg.scenes.build_scene(written)
    for key, arr in zip(("tex_diffuse", "tex_metallic_roughness", "tex_normal"), arrays):
        scene[key] = arr
    sb = pkg.capi.SceneBuffers(device, scene)
    r = pkg.capi.Renderer(device, W, H, pool_paths=P); r.bind_scene(sb)
    cam = pkg.capi.Camera(W, H); cam.set_pose(*scene["camera"])
    for _ in range(frames):
        cam.update(0.0); r.set_camera(cam.buffer); r.iterate()
    fb = r.framebuffer()
    assert np.array_equal(fb_cpp.view(np.uint32), fb.view(np.uint32))
    _, loaded = _loaded(exe, pkg, path, tmp_path, "tex.gmesh")
    loaded["materials"] = written["materials"]        # the loader assigns the layer indices while it decodes the textures (Scene.cpp:221)
    fb_orc, _ = _oracle_frames(pkg, oracle, loaded, written, W, H, P, frames, textures=arrays)
    assert np.array_equal(fb_cpp.view(np.uint32), fb_orc.view(np.uint32)), "C++ Renderer (HIP) differs from the oracle on the loaded textured scene"
    # and the textures matter: the same scene without them renders differently
    for key in ("tex_diffuse", "tex_metallic_roughness", "tex_normal"):
        scene[key] = None
    scene["materials"] = scene["materials"].copy(); scene["materials"]["textureIndices"] = -1
    sb2 = pkg.capi.SceneBuffers(device, scene)
    r2 = pkg.capi.Renderer(device, W, H, pool_paths=P); r2.bind_scene(sb2)
    cam2 = pkg.capi.Camera(W, H); cam2.set_pose(*scene["camera"])
    for _ in range(frames):
        cam2.update(0.0); r2.set_camera(cam2.buffer); r2.iterate()
    assert not np.array_equal(fb.view(np.uint32), r2.framebuffer().view(np.uint32))
    r.close(); sb.close(); r2.close(); sb2.close()


def test_cpp_row_bands_equal_the_python_split(exe, pkg):
    # host/TileGather.cpp: rowBand == tiles.py: row_bands for every frame height / rank count (bands differ by at most one row, cover the frame)
    for H, N in ((1080, 8), (1080, 7), (27, 2), (5, 8), (2160, 3), (1, 1), (64, 64)):
        out = json.loads(subprocess.run([exe, "--print-bands", str(H), str(N)], check=True, capture_output=True, text=True).stdout)
        assert [tuple(b) for b in out] == pkg.tiles.row_bands(H, N), (H, N)


@pytest.mark.gpu
def test_cpp_renderer_row_bands_equal_oracle_and_rccl_gather(exe, pkg, oracle, tmp_path):
    # the C++ host's multi-GPU path (one process per GPU; Renderer with a RowBand, TileGather over RCCL send / receive):
    #  (1) the band of every rank of a two-rank split, rendered by its own process, equals the ORACLE's render of that tile bit for bit
    #      (uneven split: 27 rows = 14 + 13; both ranks on the one GPU of the box, no gather);
    #  (2) a one-rank group goes through the whole gather path -- rendezvous file, ncclCommInitRank, the band copied into the assembled frame,
    #      the read-back -- and hands out the frame of a plain run.  Two RCCL ranks cannot share one GPU; the N-rank transport is the driver's run.
    mesh = pkg.scenes.save_gltf(pkg.scenes.random_triangles_mesh(1500, seed=9), str(tmp_path / "b.gltf"))
    path = str(tmp_path / "b.gltf")
    W, H, P, frames = 48, 27, 4096, 16
    _, loaded = _loaded(exe, pkg, path, tmp_path)
    m = dict(loaded)
    for k in ("lights", "light_count", "camera", "name"):
        m[k] = mesh[k]
    scene = pkg.scenes.build_scene(m)
    common = [exe, "--scene", path, "--size", "%dx%d" % (W, H), "--frames", str(frames), "--pool", str(P), "--live", str(P)]
    for rank, (y0, rows) in enumerate(pkg.tiles.row_bands(H, 2)):
        dump = str(tmp_path / ("band%d.f32" % rank))
        subprocess.run(common + ["--ranks", "2", "--rank", str(rank), "--device", "0", "--no-gather", "--dump", dump], check=True, cwd=str(tmp_path))
        band = np.fromfile(dump, dtype=np.float32).reshape(rows, W, 4)
        orc = oracle.Renderer(scene, W, rows, P, tile=(0, y0), threads=8)
        cam = oracle.Camera(W, H); cam.set_pose(*scene["camera"])
        for _ in range(frames):
            cam.update(); orc.set_camera(cam.buffer); orc.iterate()
        assert np.array_equal(band.view(np.uint32), orc.framebuffer().view(np.uint32)), "band %d of the C++ renderer differs from the oracle's tile" % rank
        assert int(band[..., 3].view(np.uint32).sum()) > 0
        orc.close()
    plain = str(tmp_path / "plain.f32"); full = str(tmp_path / "full.f32")
    subprocess.run(common + ["--dump", plain], check=True, cwd=str(tmp_path))
    out = subprocess.run(common + ["--ranks", "1", "--rank", "0", "--rendezvous", str(tmp_path / "rccl.id"), "--dump", full, "--pfm", str(tmp_path / "full.pfm")],
                         check=True, cwd=str(tmp_path), capture_output=True, text=True)
    assert "rank 0 of 1" in out.stdout and (tmp_path / "rccl.id").exists()
    a = np.fromfile(plain, dtype=np.float32).reshape(H, W, 4); b = np.fromfile(full, dtype=np.float32).reshape(H, W, 4)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), "the frame assembled by the one-rank RCCL gather differs from the plain run"
    raw = (tmp_path / "full.pfm").read_bytes()
    head = b"PF\n%d %d\n-1.0\n" % (W, H)
    assert raw.startswith(head) and np.array_equal(np.frombuffer(raw[len(head):], "<f4").reshape(H, W, 3)[::-1], b[..., :3])
