"""ctypes binding of oracle/liboracle.so -- the CPU oracle (test infrastructure only).

Builds the library with oracle/Makefile when it is missing or stale.  Imported by tests/, by
__graft_entry__.smoke() and by bench.py's cpu_baseline leg only.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "liboracle.so")


def build():
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("gmupt_oracle.c", "gmupt_oracle.h", "detmath.h", "sbvh_oracle.c", "Makefile")]
    srcs = [s for s in srcs if os.path.exists(s)]
    if not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in srcs):
        subprocess.run(["make", "-C", ORACLE_DIR, "-s"], check=True)
    return LIB


class CameraBuffer(C.Structure):
    _fields_ = [("pos", C.c_float * 4), ("ulc", C.c_float * 4), ("horizontal", C.c_float * 4), ("vertical", C.c_float * 4),
                ("pixelSize", C.c_float * 2), ("randomSeed", C.c_float * 2), ("envColor", C.c_float * 4),
                ("sampleCounter", C.c_int32), ("lightCount", C.c_uint32), ("sampleLights", C.c_uint32), ("pad_", C.c_uint32)]


class Scene(C.Structure):
    _fields_ = [("nodes", C.c_void_p), ("numNodes", C.c_uint32), ("tris", C.c_void_p), ("numTris", C.c_uint32),
                ("verts", C.c_void_p), ("numVerts", C.c_uint32), ("props", C.c_void_p), ("lights", C.c_void_p),
                ("materials", C.c_void_p), ("numMaterials", C.c_uint32),
                ("tex", C.c_void_p * 3), ("texSize", C.c_uint32 * 3), ("texLayers", C.c_uint32 * 3)]


class Config(C.Structure):
    _fields_ = [("poolPaths", C.c_uint32), ("livePaths", C.c_uint32), ("fbWidth", C.c_uint32), ("fbHeight", C.c_uint32),
                ("tileEnabled", C.c_uint32), ("tileX0", C.c_uint32), ("tileY0", C.c_uint32), ("pathBudget", C.c_uint32),
                ("maxDepth", C.c_uint32), ("stackSize", C.c_uint32), ("threads", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("extRays", C.c_uint64), ("extInner", C.c_uint64), ("extLeaves", C.c_uint64), ("extTris", C.c_uint64),
                ("shRays", C.c_uint64), ("shInner", C.c_uint64), ("shLeaves", C.c_uint64), ("shTris", C.c_uint64),
                ("pathsGenerated", C.c_uint64), ("pathsEnded", C.c_uint64), ("segments", C.c_uint64), ("maxStack", C.c_uint32)]


class OrcCamera(C.Structure):
    _fields_ = [("cb", CameraBuffer), ("front", C.c_float * 3), ("up", C.c_float * 3), ("left", C.c_float * 3),
                ("halfWidth", C.c_float), ("halfHeight", C.c_float), ("pitch", C.c_float), ("yaw", C.c_float),
                ("moveHysteresis", C.c_int), ("randState", C.c_uint32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.POINTER(Scene), C.POINTER(Config)]
        for name in ("orc_destroy", "orc_iterate", "orc_stage_logic", "orc_stage_new_path", "orc_stage_material_ue4",
                     "orc_stage_material_glass", "orc_stage_extension", "orc_stage_shadow", "orc_reset_stats"):
            getattr(L, name).restype = None
            getattr(L, name).argtypes = [C.c_void_p]
        L.orc_set_camera.restype = None
        L.orc_set_camera.argtypes = [C.c_void_p, C.POINTER(CameraBuffer)]
        L.orc_path_state.restype = C.POINTER(C.c_uint8); L.orc_path_state.argtypes = [C.c_void_p]
        L.orc_queues.restype = C.POINTER(C.c_uint32); L.orc_queues.argtypes = [C.c_void_p]
        L.orc_counters.restype = C.POINTER(C.c_uint32); L.orc_counters.argtypes = [C.c_void_p]
        L.orc_framebuffer.restype = C.POINTER(C.c_float); L.orc_framebuffer.argtypes = [C.c_void_p]
        L.orc_get_stats.restype = None; L.orc_get_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
        L.orc_active_paths.restype = C.c_uint32; L.orc_active_paths.argtypes = [C.c_void_p]
        L.orc_camera_init.restype = None; L.orc_camera_init.argtypes = [C.POINTER(OrcCamera), C.c_uint32, C.c_uint32]
        L.orc_camera_update_resolution.restype = None; L.orc_camera_update_resolution.argtypes = [C.POINTER(OrcCamera), C.c_uint32, C.c_uint32]
        L.orc_camera_set_pose.restype = None; L.orc_camera_set_pose.argtypes = [C.POINTER(OrcCamera)] + [C.c_float] * 5
        L.orc_camera_update.restype = None; L.orc_camera_update.argtypes = [C.POINTER(OrcCamera)]
        L.orc_msvc_rand.restype = C.c_int; L.orc_msvc_rand.argtypes = [C.POINTER(C.c_uint32)]
        L.orc_detmath_eval.restype = None; L.orc_detmath_eval.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        L.orc_debug_sample.restype = None; L.orc_debug_sample.argtypes = [C.POINTER(Scene), C.c_int, C.c_float, C.c_float, C.c_int, C.c_void_p]
        L.orc_sbvh_build.restype = C.c_int
        L.orc_sbvh_build.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int)]
        L.orc_sbvh_free.restype = None; L.orc_sbvh_free.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def detmath(fn, x, y=None):
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.zeros_like(x) if y is None else np.ascontiguousarray(y, dtype=np.float32)
    out = np.empty_like(x)
    lib().orc_detmath_eval(fn, x.ctypes.data, y.ctypes.data, out.ctypes.data, x.size)
    return out


class Camera:
    """Restatement of Source/Camera.cpp (host camera + MSVC rand seed stream)."""

    def __init__(self, width, height):
        self.c = OrcCamera()
        lib().orc_camera_init(C.byref(self.c), width, height)

    def set_pose(self, x, y, z, pitch, yaw):
        lib().orc_camera_set_pose(C.byref(self.c), x, y, z, pitch, yaw)

    def update(self):
        lib().orc_camera_update(C.byref(self.c))

    @property
    def buffer(self):
        return self.c.cb


class Renderer:
    """The oracle's Renderer: six stages in reference order under the canonical schedule."""

    def __init__(self, scene, width, height, pool, live=0, tile=None, path_budget=0, max_depth=0, stack_size=0, threads=1):
        self._keep = {k: np.ascontiguousarray(scene[k]) for k in ("nodes", "tris", "verts", "props", "lights", "materials")}
        k = self._keep
        assert k["lights"].shape[0] == 128
        self.scene = Scene(k["nodes"].ctypes.data, k["nodes"].shape[0], k["tris"].ctypes.data, k["tris"].shape[0],
                           k["verts"].ctypes.data, k["verts"].shape[0], k["props"].ctypes.data, k["lights"].ctypes.data,
                           k["materials"].ctypes.data, k["materials"].shape[0])
        for i, name in enumerate(("tex_diffuse", "tex_metallic_roughness", "tex_normal")):
            if scene.get(name) is not None:   # (layers, size, size, 4) uint8
                t = np.ascontiguousarray(scene[name], dtype=np.uint8)
                assert t.ndim == 4 and t.shape[1] == t.shape[2] and t.shape[3] == 4
                k[name] = t
                self.scene.tex[i] = t.ctypes.data; self.scene.texSize[i] = t.shape[1]; self.scene.texLayers[i] = t.shape[0]
        self.cfg = Config(pool, live, width, height, 0, 0, 0, path_budget, max_depth, stack_size, threads)
        if tile is not None:
            self.cfg.tileEnabled, self.cfg.tileX0, self.cfg.tileY0 = 1, tile[0], tile[1]
        self.h = lib().orc_create(C.byref(self.scene), C.byref(self.cfg))
        if not self.h:
            raise MemoryError("orc_create failed")
        self.pool, self.width, self.height = pool, width, height

    def sample(self, which, u, v, layer):
        out = np.zeros(4, np.float32)
        lib().orc_debug_sample(C.byref(self.scene), which, u, v, layer, out.ctypes.data)
        return out

    def set_camera(self, cb):
        buf = CameraBuffer()
        C.memmove(C.byref(buf), C.byref(cb), 112)
        lib().orc_set_camera(self.h, C.byref(buf))

    def iterate(self):
        lib().orc_iterate(self.h)

    def stage(self, name):
        getattr(lib(), "orc_stage_" + name)(self.h)

    def path_state(self):
        return np.ctypeslib.as_array(lib().orc_path_state(self.h), shape=(self.pool * 248,))

    def queues(self):
        return np.ctypeslib.as_array(lib().orc_queues(self.h), shape=(5, self.pool))

    def counters(self):
        return np.ctypeslib.as_array(lib().orc_counters(self.h), shape=(8,))

    def framebuffer(self):
        return np.ctypeslib.as_array(lib().orc_framebuffer(self.h), shape=(self.height, self.width, 4))

    def stats(self):
        s = Stats()
        lib().orc_get_stats(self.h, C.byref(s))
        return s

    def reset_stats(self):
        lib().orc_reset_stats(self.h)

    def active_paths(self):
        return lib().orc_active_paths(self.h)

    def close(self):
        if self.h:
            lib().orc_destroy(self.h)
            self.h = None


# byte offsets (per path, x pool) of the reference path-state fields, Assets/Shaders/structs.h:19-48
STATE_FIELDS = {
    "rayOrigin": (0, 16, 3, "f"), "rayDirection": (16, 16, 3, "f"), "matColor": (32, 16, 3, "f"), "matMR": (48, 8, 2, "f"),
    "normal": (56, 16, 3, "f"), "surfacePoint": (72, 16, 3, "f"), "baryCoord": (88, 16, 3, "f"), "hitDistance": (104, 4, 1, "f"),
    "triangle": (108, 16, 4, "u"), "shadowrayOrigin": (124, 16, 3, "f"), "shadowrayDirection": (140, 16, 3, "f"),
    "lightIndex": (156, 4, 1, "u"), "lightDistance": (160, 4, 1, "f"), "inShadow": (164, 4, 1, "u"), "radiance": (168, 16, 3, "f"),
    "throughput": (184, 16, 3, "f"), "lightThroughput": (200, 16, 3, "f"), "directLight": (216, 16, 3, "f"),
    "pathLength": (232, 4, 1, "u"), "screenCoord": (236, 8, 2, "u"), "isEmitter": (244, 4, 1, "u"),
}


def state_field(raw, pool, name, live=None):
    """View of one field of a 248-byte-per-path reference state buffer as uint32 (bit patterns), shape (n, comps)."""
    off, slot, comps, _ = STATE_FIELDS[name]
    n = pool if live is None else live
    words = np.frombuffer(raw, dtype=np.uint32)
    base = off * pool // 4
    return words[base: base + (slot // 4) * pool].reshape(pool, slot // 4)[:n, :comps]


def sbvh_build(verts, indices, vertex_material=None):
    """The oracle's restatement of the reference SBVH build + flatten: (nodes, tris) as raw 48-byte / 16-byte records."""
    verts = np.ascontiguousarray(verts, np.float32); indices = np.ascontiguousarray(indices, np.int32)
    vm = None if vertex_material is None else np.ascontiguousarray(vertex_material, np.uint32)
    pn, pt, nref = C.c_void_p(), C.c_void_p(), C.c_int(0)
    n = lib().orc_sbvh_build(verts.ctypes.data, verts.shape[0], indices.ctypes.data, indices.shape[0],
                             vm.ctypes.data if vm is not None else None, C.byref(pn), C.byref(pt), C.byref(nref))
    nodes = np.frombuffer(C.string_at(pn, n * 48), dtype=np.uint8).copy()
    tris = np.frombuffer(C.string_at(pt, nref.value * 16), dtype=np.uint8).copy()
    lib().orc_sbvh_free(pn); lib().orc_sbvh_free(pt)
    return nodes, tris, n, nref.value
