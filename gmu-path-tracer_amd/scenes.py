"""Synthetic scene generators for the BASELINE.json configurations.

The reference's own assets are unresolved Git-LFS pointers (Assets/Models/*.gltf, *.bin), so the parity and
bench scenes are generated here with fixed seeds.  A scene is a dict of numpy arrays laid out exactly as the
reference's BVHWrapper produces them (Source/BVHWrapper.cpp:13-96): world-space float3 vertices, one
TriangleProperties per VERTEX (smooth normal, uv, material id), triangle vertex indices, 48-byte materials,
32-byte lights (128-entry table), plus the camera pose of the scene's .params row 0.
"""
import numpy as np

from . import capi


def _material(color, metallic=0.0, roughness=1.0, mtype=capi.MATERIAL_UE4):
    m = np.zeros((), dtype=capi.material_dtype)
    m["color"] = (color[0], color[1], color[2], 1.0)
    m["metallic"], m["roughness"] = metallic, roughness
    m["refractIndex"], m["transmittance"] = 1.458, 0.0
    m["textureIndices"] = (-1, -1, -1)
    m["materialType"] = mtype
    return m


def _lights(rows):
    """rows: (x, y, z, falloff, r, g, b, radius) as in a .params file (Source/Scene.cpp:43-54)."""
    out = np.zeros(capi.MAX_LIGHTS, dtype=capi.light_dtype)
    for i, r in enumerate(rows):
        out[i]["position"] = r[0:3]
        out[i]["falloff"] = r[3]
        out[i]["emission"] = r[4:7]
        out[i]["radius"] = r[7]
    return out


DEFAULT_LIGHTS = [(13.0, 4.5, 4.5, 100.0, 80.0, 80.0, 40.0, 0.5), (0.0, 4.5, 2.0, 100.0, 80.0, 80.0, 40.0, 0.5)]  # Scene.cpp:60-61
DEFAULT_CAMERA = (1.0, 3.0, 8.0, 0.0, 270.0)  # Scene.cpp:59


class _MeshBuilder:
    """Accumulates vertices / normals / materials / triangles as a list of numpy chunks (10 M-triangle scenes stay cheap)."""

    def __init__(self):
        self.v, self.n, self.m, self.t = [], [], [], []
        self.count = 0

    def _add(self, verts, normals, mats, tris):
        verts = np.asarray(verts, np.float64).reshape(-1, 3)
        self.v.append(verts); self.n.append(np.asarray(normals, np.float64).reshape(-1, 3))
        self.m.append(np.asarray(mats, np.uint32).reshape(-1)); self.t.append(np.asarray(tris, np.int64).reshape(-1, 3) + self.count)
        self.count += verts.shape[0]

    def quad(self, p0, p1, p2, p3, normal, mat):
        self._add([p0, p1, p2, p3], [normal] * 4, [mat] * 4, [(0, 1, 2), (0, 2, 3)])

    def box(self, center, half, yaw_deg, mat):
        c = np.asarray(center, np.float64); h = np.asarray(half, np.float64)
        a = np.deg2rad(yaw_deg); R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
        def P(sx, sy, sz):
            return tuple(c + R @ (h * np.array([sx, sy, sz])))
        faces = [((1, 0, 0), [(1, -1, -1), (1, 1, -1), (1, 1, 1), (1, -1, 1)]), ((-1, 0, 0), [(-1, -1, 1), (-1, 1, 1), (-1, 1, -1), (-1, -1, -1)]),
                 ((0, 1, 0), [(-1, 1, -1), (-1, 1, 1), (1, 1, 1), (1, 1, -1)]), ((0, -1, 0), [(-1, -1, 1), (-1, -1, -1), (1, -1, -1), (1, -1, 1)]),
                 ((0, 0, 1), [(-1, -1, 1), (1, -1, 1), (1, 1, 1), (-1, 1, 1)]), ((0, 0, -1), [(1, -1, -1), (-1, -1, -1), (-1, 1, -1), (1, 1, -1)])]
        for nrm, corners in faces:
            n = tuple(R @ np.array(nrm, np.float64))
            self.quad(*[P(*k) for k in corners], n, mat)

    def mesh(self, verts, normals, tris, mat):
        self._add(verts, normals, np.full(len(verts), mat, np.uint32), tris)

    def arrays(self):
        return (np.concatenate(self.v).astype(np.float32), np.concatenate(self.n).astype(np.float32),
                np.concatenate(self.m).astype(np.uint32), np.concatenate(self.t).astype(np.int32))


def _room(mb, lo, hi, mats, open_front=False):
    """axis-aligned room with inward normals; mats = (floor, ceiling, back, left, right, front)"""
    x0, y0, z0 = lo; x1, y1, z1 = hi
    mb.quad((x0, y0, z1), (x1, y0, z1), (x1, y0, z0), (x0, y0, z0), (0, 1, 0), mats[0])
    mb.quad((x0, y1, z0), (x1, y1, z0), (x1, y1, z1), (x0, y1, z1), (0, -1, 0), mats[1])
    mb.quad((x0, y0, z0), (x1, y0, z0), (x1, y1, z0), (x0, y1, z0), (0, 0, 1), mats[2])
    mb.quad((x0, y0, z1), (x0, y0, z0), (x0, y1, z0), (x0, y1, z1), (1, 0, 0), mats[3])
    mb.quad((x1, y0, z0), (x1, y0, z1), (x1, y1, z1), (x1, y1, z0), (-1, 0, 0), mats[4])
    if not open_front:
        mb.quad((x1, y0, z1), (x0, y0, z1), (x0, y1, z1), (x1, y1, z1), (0, 0, -1), mats[5])


def cornell_mesh():
    """Config 2: 5 walls (10 tris) + 2 boxes (24 tris) = 34 triangles, room 10x10x10 centred at (0,5,0), diffuse only."""
    mb = _MeshBuilder()
    _room(mb, (-5, 0, -5), (5, 10, 5), (0, 0, 0, 1, 2, 0), open_front=True)
    mb.box((-1.8, 3.0, -1.5), (1.5, 3.0, 1.5), 18.0, 0)   # tall box, rotated
    mb.box((1.7, 1.5, 1.0), (1.5, 1.5, 1.5), -17.0, 0)    # short box
    v, n, m, t = mb.arrays()
    materials = np.stack([_material((0.73, 0.73, 0.73)), _material((0.65, 0.05, 0.05)), _material((0.12, 0.45, 0.15))])
    return {"verts": v, "normals": n, "vertex_material": m, "indices": t, "materials": materials,
            "lights": _lights(DEFAULT_LIGHTS), "light_count": 2, "camera": DEFAULT_CAMERA, "name": "cornell34"}


def icosphere(subdiv):
    t = (1.0 + np.sqrt(5.0)) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t), (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    v = [tuple(np.asarray(p, np.float64) / np.linalg.norm(p)) for p in v]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
         (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    for _ in range(subdiv):
        cache, nf = {}, []
        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                p = (np.asarray(v[a]) + np.asarray(v[b])) * 0.5
                v.append(tuple(p / np.linalg.norm(p)))
                cache[key] = len(v) - 1
            return cache[key]
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    return np.asarray(v, np.float64), np.asarray(f, np.int32)


def spheres_mesh(n_spheres=202, subdiv=3, seed=1234, floor_quads=20, half_extent=12.0, height=10.0):
    """Configs 3/4 (202 icospheres at subdivision 3 -> 259 372 triangles) and config 5 (1953 at subdivision 4).

    Closed room (12 tris) + icospheres with shared vertices on a jittered grid, radius 0.3-0.8, + a floor_quads^2 quad
    floor.  Materials: 60 % rough dielectric, 25 % metallic (roughness 0.1-0.6), 15 % glass; two sphere lights.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    mb = _MeshBuilder()
    E, H = half_extent, height
    _room(mb, (-E, -0.05, -E), (E, H, E), (0, 0, 0, 1, 2, 0), open_front=False)
    materials = [_material((0.7, 0.7, 0.7)), _material((0.6, 0.1, 0.1)), _material((0.1, 0.5, 0.15))]
    # tessellated floor
    fq = floor_quads
    xs = np.linspace(-E, E, fq + 1)
    for i in range(fq):
        for j in range(fq):
            mb.quad((xs[i], 0.0, xs[j + 1]), (xs[i + 1], 0.0, xs[j + 1]), (xs[i + 1], 0.0, xs[j]), (xs[i], 0.0, xs[j]), (0, 1, 0), 0)
    sv, sf = icosphere(subdiv)
    g = max(1, int(np.ceil(np.sqrt(n_spheres))))
    cell = 2.0 * (E - 1.0) / g
    cells = rng.permutation(g * g)[:n_spheres]
    for k, cidx in enumerate(cells):
        ci, cj = divmod(int(cidx), g)
        radius = rng.uniform(0.3, 0.8) * min(1.0, cell / 1.7)
        cx = -(E - 1.0) + (ci + 0.5) * cell + rng.uniform(-0.25, 0.25) * (cell - 2 * radius)
        cz = -(E - 1.0) + (cj + 0.5) * cell + rng.uniform(-0.25, 0.25) * (cell - 2 * radius)
        cy = rng.uniform(radius + 0.02, 6.0)
        u = rng.uniform()
        col = tuple(rng.uniform(0.25, 0.9, size=3))
        if u < 0.60:
            mat = _material(col, 0.0, rng.uniform(0.5, 1.0))
        elif u < 0.85:
            mat = _material(col, 1.0, rng.uniform(0.1, 0.6))
        else:
            mat = _material((0.95, 0.95, 0.95), 0.0, 0.1, capi.MATERIAL_GLASS)
        if len(materials) < capi.MAX_LIGHTS:  # the material cbuffer holds 128 entries (logic.hlsl:8)
            materials.append(mat); mid = len(materials) - 1
        else:
            mid = 3 + (k % (capi.MAX_LIGHTS - 3))
        mb.mesh(sv * radius + np.array([cx, cy, cz]), sv, sf, mid)
    v, n, m, t = mb.arrays()
    lights = _lights([(-5.0, 8.0, 3.0, 100.0, 80.0, 80.0, 40.0, 0.5), (5.0, 8.0, -3.0, 100.0, 80.0, 80.0, 40.0, 0.5)])
    return {"verts": v, "normals": n, "vertex_material": m, "indices": t, "materials": np.stack(materials),
            "lights": lights, "light_count": 2, "camera": (0.0, 4.0, 11.0, -8.0, 270.0), "name": "spheres%d_s%d" % (n_spheres, subdiv)}


def random_triangles_mesh(n=2000, seed=1, extent=10.0, size=1.0):
    """Unstructured triangle soup (builder / traversal tests)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    c = rng.uniform(-extent, extent, size=(n, 1, 3))
    v = (c + rng.uniform(-size, size, size=(n, 3, 3))).reshape(-1, 3).astype(np.float32)
    t = np.arange(3 * n, dtype=np.int32).reshape(n, 3)
    e1 = v[t[:, 1]] - v[t[:, 0]]; e2 = v[t[:, 2]] - v[t[:, 0]]
    fn = np.cross(e1, e2); fn /= np.maximum(np.linalg.norm(fn, axis=1, keepdims=True), 1e-20)
    nrm = np.repeat(fn, 3, axis=0).astype(np.float32)
    mats = np.stack([_material((0.7, 0.7, 0.7)), _material((0.8, 0.6, 0.2), 1.0, 0.3), _material((0.95, 0.95, 0.95), 0.0, 0.1, capi.MATERIAL_GLASS)])
    vm = np.repeat(rng.integers(0, 3, size=n).astype(np.uint32), 3)
    return {"verts": v, "normals": nrm, "vertex_material": vm, "indices": t, "materials": mats,
            "lights": _lights([(0.0, 14.0, 0.0, 100.0, 80.0, 80.0, 40.0, 0.5), (6.0, 3.0, 6.0, 100.0, 40.0, 80.0, 80.0, 0.7)]),
            "light_count": 2, "camera": (0.0, 2.0, 24.0, 0.0, 270.0), "name": "soup%d" % n}


def deep_chain_mesh(n=100, factor=0.6, per=2, seed=0):
    """Geometrically shrinking triangle clusters nested towards a corner: an unbalanced SAH tree of depth ~30, used to exercise
    traversal stacks deeper than the LDS portion (the reference's 16-entry stack would overflow here, quirk Q23)."""
    rng = np.random.default_rng(seed)
    v = []
    for k in range(n):
        s = factor ** k * 10.0
        c = np.array([s, s * 0.5, s * 0.3])
        for _ in range(per):
            v.append(c + rng.uniform(-0.2, 0.2, (3, 3)) * s)
    v = np.concatenate(v).astype(np.float32)
    t = np.arange(len(v), dtype=np.int32).reshape(-1, 3)
    e1 = v[t[:, 1]] - v[t[:, 0]]; e2 = v[t[:, 2]] - v[t[:, 0]]
    fn = np.cross(e1, e2); fn /= np.maximum(np.linalg.norm(fn, axis=1, keepdims=True), 1e-30)
    mats = np.stack([_material((0.8, 0.8, 0.8)), _material((0.9, 0.7, 0.3), 1.0, 0.2)])
    vm = np.repeat((np.arange(len(t)) % 2).astype(np.uint32), 3)
    return {"verts": v, "normals": np.repeat(fn, 3, axis=0).astype(np.float32), "vertex_material": vm, "indices": t, "materials": mats,
            "lights": _lights([(4.0, 6.0, 4.0, 100.0, 80.0, 80.0, 40.0, 0.5), (-2.0, 3.0, 5.0, 100.0, 40.0, 80.0, 80.0, 0.5)]),
            "light_count": 2, "camera": (0.0, 0.0, 0.0, 25.6, 16.7), "name": "chain%d" % n}  # at the apex, looking along the chain


def textured_mesh(seed=5):
    """Cornell-like room whose walls and boxes use all three texture slots (base colour, metallic/roughness, normal map) --
    scene for the texture rows of logic.hlsl:99-124.  Planar UVs (tiling beyond [0,1] exercises WRAP), 3 layers of 16x16 texels."""
    rng = np.random.default_rng(seed)
    m = cornell_mesh()
    v = m["verts"]
    uv = np.stack([v[:, 0] * 0.37 + v[:, 1] * 0.11, v[:, 2] * 0.29 - v[:, 1] * 0.23], axis=1).astype(np.float32)
    size, layers = 16, 3
    yy, xx = np.mgrid[0:size, 0:size]
    diffuse = np.zeros((layers, size, size, 4), np.uint8)
    diffuse[0, ..., :3] = np.where(((xx // 4 + yy // 4) % 2)[..., None] == 0, (230, 230, 230), (40, 60, 200)); diffuse[0, ..., 3] = 255
    diffuse[1] = rng.integers(30, 255, (size, size, 4)); diffuse[2, ..., 0] = 8 * xx + 60; diffuse[2, ..., 1] = 200; diffuse[2, ..., 2] = 8 * yy + 40; diffuse[2, ..., 3] = 255
    mr = np.zeros((layers, size, size, 4), np.uint8)
    mr[0, ..., 0] = np.where((xx % 8) < 4, 255, 0); mr[0, ..., 1] = 64 + 8 * yy       # .x metallic stripes, .y roughness ramp (quirk Q11)
    mr[1] = rng.integers(0, 255, (size, size, 4)); mr[2, ..., 0] = 0; mr[2, ..., 1] = 230
    nrm = np.zeros((layers, size, size, 4), np.uint8)
    ang = rng.uniform(0, 2 * np.pi, (layers, size, size)); tilt = rng.uniform(0, 0.45, (layers, size, size))
    nrm[..., 0] = np.clip((np.cos(ang) * tilt * 0.5 + 0.5) * 255, 0, 255); nrm[..., 1] = np.clip((np.sin(ang) * tilt * 0.5 + 0.5) * 255, 0, 255)
    nrm[..., 2] = np.clip((np.sqrt(1 - tilt ** 2) * 0.5 + 0.5) * 255, 0, 255); nrm[..., 3] = 255
    mats = m["materials"].copy()
    mats[0]["textureIndices"] = (0, 0, 1)      # grey walls: checker colour, metallic stripes, bumpy
    mats[1]["textureIndices"] = (1, -1, -1)    # red wall: colour only
    mats[2]["textureIndices"] = (-1, 2, 2)     # green wall: rough dielectric + normal map, material colour
    m.update({"uv": uv, "materials": mats, "tex_diffuse": diffuse, "tex_metallic_roughness": mr, "tex_normal": nrm, "name": "cornell_textured"})
    return m


def build_scene(mesh, sbvh_params=None):
    """BVHWrapper::buildSBVH: host SBVH build + flatten -> the buffers Renderer::draw binds (t0-t4, b1)."""
    built = capi.sbvh_build(mesh["verts"], mesh["indices"], mesh["vertex_material"], sbvh_params)
    nv = mesh["verts"].shape[0]
    props = np.zeros(nv, dtype=capi.tri_props_dtype)
    props["normal"] = mesh["normals"]
    props["materialID"] = mesh["vertex_material"]
    if "uv" in mesh:
        props["uv"] = mesh["uv"]
    scene = {"nodes": built["nodes"], "tris": built["tris"], "verts": np.ascontiguousarray(mesh["verts"], np.float32),
             "props": props, "lights": mesh["lights"], "materials": np.ascontiguousarray(mesh["materials"]),
             "light_count": mesh["light_count"], "camera": mesh["camera"], "name": mesh["name"],
             "sah": built["sah"], "depth": built["depth"], "ref_triangle": built["ref_triangle"],
             "num_triangles": int(mesh["indices"].shape[0])}
    for k in ("tex_diffuse", "tex_metallic_roughness", "tex_normal"):
        scene[k] = mesh.get(k)
    return scene


def save_gmesh(mesh, path, params_path=None):
    """Writes the ".gmesh" dump host/MeshData.cpp reads, and optionally the sibling ".params" CSV in the reference's format
    (row 0: camera x,y,z,pitch,yaw; rows 1..: light x,y,z,falloff,r,g,b,radius -- Source/Scene.cpp:34-55)."""
    nv, nt, nm = mesh["verts"].shape[0], mesh["indices"].shape[0], mesh["materials"].shape[0]
    with open(path, "wb") as f:
        f.write(b"GMESH001")
        f.write(np.array([nv, nt, nm, 1 if "uv" in mesh else 0], np.uint32).tobytes())
        f.write(np.ascontiguousarray(mesh["verts"], np.float32).tobytes())
        f.write(np.ascontiguousarray(mesh["normals"], np.float32).tobytes())
        if "uv" in mesh:
            f.write(np.ascontiguousarray(mesh["uv"], np.float32).tobytes())
        f.write(np.ascontiguousarray(mesh["vertex_material"], np.uint32).tobytes())
        f.write(np.ascontiguousarray(mesh["indices"], np.int32).tobytes())
        f.write(np.ascontiguousarray(mesh["materials"]).tobytes())
    if params_path:
        with open(params_path, "w") as f:
            f.write(", ".join(repr(float(v)) for v in mesh["camera"]) + "\n")
            for i in range(mesh["light_count"]):
                L = mesh["lights"][i]
                row = list(L["position"]) + [L["falloff"]] + list(L["emission"]) + [L["radius"]]
                f.write(", ".join(repr(float(v)) for v in row) + "\n")


def load_gmesh(path):
    """Reads a ".gmesh" dump (save_gmesh above, or gmupt_render --dump-mesh = what the C++ scene loader produced) as a mesh dict
    without camera / lights."""
    raw = open(path, "rb").read()
    assert raw[:8] == b"GMESH001"
    nv, nt, nm, has_uv = np.frombuffer(raw, np.uint32, 4, 8)
    off = 24
    def take(dtype, count):
        nonlocal off
        a = np.frombuffer(raw, dtype, count, off).copy()
        off += a.nbytes
        return a
    mesh = {"verts": take(np.float32, nv * 3).reshape(-1, 3), "normals": take(np.float32, nv * 3).reshape(-1, 3)}
    if has_uv:
        mesh["uv"] = take(np.float32, nv * 2).reshape(-1, 2)
    mesh["vertex_material"] = take(np.uint32, nv)
    mesh["indices"] = take(np.int32, nt * 3).reshape(-1, 3)
    mesh["materials"] = take(capi.material_dtype, nm)
    return mesh


def encode_png_rgba8(rgba):
    """Minimal PNG writer (RGBA, 8 bit, filter 0, zlib) for the texture files of save_gltf."""
    import struct, zlib
    rgba = np.ascontiguousarray(rgba, np.uint8)
    h, w = rgba.shape[:2]
    raw = b"".join(b"\x00" + rgba[y].tobytes() for y in range(h))

    def chunk(kind, body):
        return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body) & 0xFFFFFFFF)
    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b"")


_TEX_KEYS = ("tex_diffuse", "tex_metallic_roughness", "tex_normal")
_GLTF_TEX_SLOTS = (("pbrMetallicRoughness", "baseColorTexture"), ("pbrMetallicRoughness", "metallicRoughnessTexture"), (None, "normalTexture"))


def save_gltf(mesh, path, node_transform=None, with_normals=True, texture_scale=None):
    """Writes `mesh` as glTF 2.0 (.gltf + .bin next to it), one TRIANGLES primitive per material, and returns the mesh in the vertex
    order it was written (what a loader reproduces).  UVs are stored un-flipped (v -> 1 - v), as glTF files are before the reference's
    aiProcess_FlipUVs.  Optionally writes a sibling .params file (camera + lights) like save_gmesh."""
    import json as _json
    tri_mat = mesh["vertex_material"][mesh["indices"][:, 0]]
    verts, normals, uvs, vmat, tris = [], [], [], [], []
    prims, accessors, views, blob = [], [], [], bytearray()

    def add_view(data, target=None):
        while len(blob) % 4:
            blob.append(0)
        views.append({"buffer": 0, "byteOffset": len(blob), "byteLength": len(data)})
        if target:
            views[-1]["target"] = target
        blob.extend(data)
        return len(views) - 1

    base = 0
    for m in range(mesh["materials"].shape[0]):
        sel = np.where(tri_mat == m)[0]
        if sel.size == 0:
            continue
        used, inv = np.unique(mesh["indices"][sel].reshape(-1), return_inverse=True)
        p = mesh["verts"][used].astype(np.float32); n = mesh["normals"][used].astype(np.float32)
        uv = (mesh["uv"][used] if "uv" in mesh else np.zeros((used.size, 2))).astype(np.float32)
        idx = inv.reshape(-1, 3).astype(np.uint32)
        attrs = {}
        accessors.append({"bufferView": add_view(p.tobytes(), 34962), "componentType": 5126, "count": int(used.size), "type": "VEC3",
                          "min": p.min(0).tolist(), "max": p.max(0).tolist()}); attrs["POSITION"] = len(accessors) - 1
        if with_normals:
            accessors.append({"bufferView": add_view(n.tobytes(), 34962), "componentType": 5126, "count": int(used.size), "type": "VEC3"}); attrs["NORMAL"] = len(accessors) - 1
        uv_file = uv.copy(); uv_file[:, 1] = 1.0 - uv_file[:, 1]
        accessors.append({"bufferView": add_view(uv_file.astype(np.float32).tobytes(), 34962), "componentType": 5126, "count": int(used.size), "type": "VEC2"}); attrs["TEXCOORD_0"] = len(accessors) - 1
        accessors.append({"bufferView": add_view(idx.tobytes(), 34963), "componentType": 5125, "count": int(idx.size), "type": "SCALAR"})
        prims.append({"attributes": attrs, "indices": len(accessors) - 1, "material": m, "mode": 4})
        verts.append(p); normals.append(n); uvs.append((1.0 - uv_file[:, 1:2]) * 1.0); vmat.append(np.full(used.size, m, np.uint32)); tris.append(idx.astype(np.int32) + base)
        uvs[-1] = np.concatenate([uv_file[:, 0:1], 1.0 - uv_file[:, 1:2]], axis=1).astype(np.float32)
        base += used.size
    mats = []
    for mm in mesh["materials"]:
        d = {"pbrMetallicRoughness": {"baseColorFactor": [float(v) for v in mm["color"]], "metallicFactor": float(mm["metallic"]), "roughnessFactor": float(mm["roughness"])}}
        if int(mm["materialType"]) == capi.MATERIAL_GLASS:
            d["alphaMode"] = "BLEND"
        mats.append(d)
    # textures: one PNG file per (material, slot) that uses a layer; the loader numbers the layers in material order (Scene.cpp:221).
    # texture_scale = {(material, slot): k} writes that image k times larger (nearest), so that the loader has to resize it.
    images, textures, tex_files = [], [], [[], [], []]
    stem = path.rsplit("/", 1)[-1].rsplit(".", 1)[0]
    folder = path.rsplit("/", 1)[0] + "/" if "/" in path else ""
    new_materials = np.array(mesh["materials"], copy=True)
    for i, mm in enumerate(mesh["materials"]):
        for t, key in enumerate(_TEX_KEYS):
            layer = int(mm["textureIndices"][t])
            if layer < 0 or mesh.get(key) is None:
                new_materials[i]["textureIndices"][t] = -1
                continue
            img = mesh[key][layer]
            k = (texture_scale or {}).get((i, t), 1)
            if k != 1:
                img = np.kron(img, np.ones((k, k, 1), np.uint8))
            data = encode_png_rgba8(img)
            name = "%s_m%d_t%d.png" % (stem, i, t)
            with open(folder + name, "wb") as f:
                f.write(data)
            images.append({"uri": name}); textures.append({"source": len(images) - 1})
            group, slot = _GLTF_TEX_SLOTS[t]
            (mats[i][group] if group else mats[i])[slot] = {"index": len(textures) - 1}
            new_materials[i]["textureIndices"][t] = len(tex_files[t])
            tex_files[t].append(data)
    node = {"mesh": 0}
    if node_transform:
        node.update(node_transform)
    doc = {"asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": [0]}], "nodes": [node], "meshes": [{"primitives": prims}], "materials": mats,
           **({"images": images, "textures": textures} if images else {}),
           "accessors": accessors, "bufferViews": views, "buffers": [{"uri": path.rsplit("/", 1)[-1].rsplit(".", 1)[0] + ".bin", "byteLength": len(blob)}]}
    with open(path, "w") as f:
        _json.dump(doc, f)
    with open(path.rsplit(".", 1)[0] + ".bin", "wb") as f:
        f.write(bytes(blob))
    with open(path.rsplit(".", 1)[0] + ".params", "w") as f:
        f.write(", ".join(repr(float(v)) for v in mesh["camera"]) + "\n")
        for i in range(mesh["light_count"]):
            L = mesh["lights"][i]
            f.write(", ".join(repr(float(v)) for v in list(L["position"]) + [L["falloff"]] + list(L["emission"]) + [L["radius"]]) + "\n")
    out = dict(mesh)
    out["materials"] = new_materials
    out["texture_files"] = tex_files  # encoded PNGs per slot in layer order: capi.texture_array_from_png() turns them into the arrays
    for key in _TEX_KEYS:
        out.pop(key, None)
    out.update({"verts": np.concatenate(verts), "normals": np.concatenate(normals), "uv": np.concatenate(uvs), "vertex_material": np.concatenate(vmat),
                "indices": np.concatenate(tris)})
    return out


def gltf_to_glb(gltf_path, glb_path):
    """Repacks a .gltf written by save_gltf (external .bin, external PNG images) as one binary .glb: the buffer becomes the BIN chunk,
    the images become bufferViews of it.  The sibling .params file is copied."""
    import json as _json, os as _os, struct as _struct, shutil as _shutil
    folder = _os.path.dirname(gltf_path)
    doc = _json.load(open(gltf_path))
    blob = bytearray(open(_os.path.join(folder, doc["buffers"][0]["uri"]), "rb").read())
    for img in doc.get("images", []):
        data = open(_os.path.join(folder, img.pop("uri")), "rb").read()
        while len(blob) % 4:
            blob.append(0)
        doc["bufferViews"].append({"buffer": 0, "byteOffset": len(blob), "byteLength": len(data)})
        img["bufferView"] = len(doc["bufferViews"]) - 1; img["mimeType"] = "image/png"
        blob.extend(data)
    while len(blob) % 4:
        blob.append(0)
    doc["buffers"] = [{"byteLength": len(blob)}]
    js = _json.dumps(doc).encode()
    js += b" " * (-len(js) % 4)
    total = 12 + 8 + len(js) + 8 + len(blob)
    with open(glb_path, "wb") as f:
        f.write(_struct.pack("<III", 0x46546C67, 2, total))
        f.write(_struct.pack("<II", len(js), 0x4E4F534A)); f.write(js)
        f.write(_struct.pack("<II", len(blob), 0x004E4942)); f.write(bytes(blob))
    src = gltf_path.rsplit(".", 1)[0] + ".params"
    if _os.path.exists(src):
        _shutil.copyfile(src, glb_path.rsplit(".", 1)[0] + ".params")
