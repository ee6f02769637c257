"""ctypes binding of libgmupt.so -- the binding a Python host would add on top of include/gmupt.h.

This is plumbing only: every call goes straight to the C-ABI, there is no CPU fallback.  If the HIP
library is missing or a HIP call fails, GmuptError is raised with gmupt_last_error().
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

MAX_LIGHTS = 128
PATHCOUNT = 1 << 21
REF_GRID_THREADS = 34 * 8 * 256
STATE_BYTES = 248

(BUFFER_BVH_NODES, BUFFER_TRIANGLES, BUFFER_VERTICES, BUFFER_LIGHTS, BUFFER_TRI_PROPS, BUFFER_MATERIALS, BUFFER_TEXTURE_ARRAY) = range(7)
STAGE_SHADE, STAGE_EXTEND, STAGE_SHADOW, STAGE_RAYCASTS = range(4)
MATERIAL_UE4, MATERIAL_GLASS = 0, 1

# numpy views of the reference PODs (sizes asserted below)
bvh_node_dtype = np.dtype([("min", "<f4", 3), ("pad0", "<f4"), ("max", "<f4", 3), ("pad1", "<f4"),
                           ("left", "<i4"), ("right", "<i4"), ("isLeaf", "<i4"), ("pad2", "<f4")])
triangle_dtype = np.dtype([("v", "<i4", 3), ("materialID", "<u4")])
tri_props_dtype = np.dtype([("normal", "<f4", 3), ("pad0", "<f4"), ("uv", "<f4", 2), ("materialID", "<u4"), ("pad1", "<f4")])
light_dtype = np.dtype([("position", "<f4", 3), ("falloff", "<f4"), ("emission", "<f4", 3), ("radius", "<f4")])
material_dtype = np.dtype([("color", "<f4", 4), ("metallic", "<f4"), ("roughness", "<f4"), ("refractIndex", "<f4"),
                           ("transmittance", "<f4"), ("textureIndices", "<i4", 3), ("materialType", "<u4")])
assert bvh_node_dtype.itemsize == 48 and triangle_dtype.itemsize == 16 and tri_props_dtype.itemsize == 32
assert light_dtype.itemsize == 32 and material_dtype.itemsize == 48


class CameraBuffer(C.Structure):
    _fields_ = [("position", C.c_float * 4), ("upperLeftCorner", C.c_float * 4), ("horizontal", C.c_float * 4),
                ("vertical", C.c_float * 4), ("pixelSize", C.c_float * 2), ("randomSeed", C.c_float * 2),
                ("envColor", C.c_float * 4), ("iterationCounter", C.c_int32), ("lightCount", C.c_uint32),
                ("sampleLights", C.c_uint32), ("pad_", C.c_uint32)]


assert C.sizeof(CameraBuffer) == 112


class RendererDesc(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("pool_paths", C.c_uint32), ("live_paths", C.c_uint32),
                ("tile_enabled", C.c_uint32), ("tile_x0", C.c_uint32), ("tile_y0", C.c_uint32),
                ("path_budget", C.c_uint32), ("max_depth", C.c_uint32), ("collect_stats", C.c_uint32)]


STAT_STACK_OVERFLOW, STAT_FUSED_CAST, STAT_CAST_FETCH, STAT_STACK_SPILL, STAT_CAST_ABORTED, STAT_CAST_WIDE = 1, 2, 4, 8, 16, 32


class Stats(C.Structure):
    _fields_ = [("iterations", C.c_uint64), ("paths_generated", C.c_uint64), ("paths_completed", C.c_uint64),
                ("segments", C.c_uint64), ("active_paths", C.c_uint32), ("flags", C.c_uint32),
                ("ext_rays", C.c_uint64), ("ext_inner", C.c_uint64), ("ext_leaves", C.c_uint64), ("ext_tris", C.c_uint64),
                ("sh_rays", C.c_uint64), ("sh_inner", C.c_uint64), ("sh_leaves", C.c_uint64), ("sh_tris", C.c_uint64),
                ("ms_logic", C.c_double), ("ms_scan", C.c_double), ("ms_accumulate", C.c_double), ("ms_material", C.c_double),
                ("ms_extend", C.c_double), ("ms_shadow", C.c_double), ("timed_iterations", C.c_uint64),
                ("ext_wave_inner", C.c_uint64), ("ext_wave_tris", C.c_uint64), ("sh_wave_inner", C.c_uint64), ("sh_wave_tris", C.c_uint64), ("ext_depth_hist", C.c_uint64 * 32),
                ("lane_census", C.c_uint64 * 4), ("cast_waves", C.c_uint64), ("cast_wave_ticks", C.c_uint64), ("cast_wave_ticks_max", C.c_uint64),
                ("cast_drain_ticks", C.c_uint64), ("cast_drain_iters", C.c_uint64), ("cast_drain_busy_lanes", C.c_uint64),
                ("cast_wave_end_hist", C.c_uint64 * 32), ("ray_inner_hist", C.c_uint64 * 32),
                ("ext_top_inner", C.c_uint64), ("sh_top_inner", C.c_uint64), ("cast_helper_subtrees", C.c_uint64),
                ("cast_nested_helpers", C.c_uint64), ("cast_redo_rays", C.c_uint64), ("wide_nodes", C.c_uint64), ("wide_top_nodes", C.c_uint64), ("wide_stack_bound", C.c_uint64), ("wide_pairs", C.c_uint64), ("wide_pair_fetches", C.c_uint64), ("wide_box_tests", C.c_uint64),
                ("wide_iterations", C.c_uint64), ("wide_general_iterations", C.c_uint64)]

    def as_dict(self):
        return {n: (list(getattr(self, n)) if hasattr(getattr(self, n), "__len__") else getattr(self, n)) for n, _ in self._fields_}


class SbvhParams(C.Structure):
    _fields_ = [("split_alpha", C.c_float), ("max_depth", C.c_int32), ("max_spatial_depth", C.c_int32),
                ("min_leaf_size", C.c_int32), ("max_leaf_size", C.c_int32), ("node_cost", C.c_float), ("tri_cost", C.c_float)]


ERR_CAST_FAULT = -7


class GmuptError(RuntimeError):
    def __init__(self, msg, code=0):
        super().__init__(msg)
        self.code = code


# every symbol include/gmupt.h declares: name -> (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "gmupt_device_create": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "gmupt_device_destroy": (None, [_P]),
    "gmupt_last_error": (C.c_char_p, []),
    "gmupt_device_count": (C.c_int, []),
    "gmupt_buffer_create": (C.c_int, [_P, C.c_int, _P, C.c_size_t, C.POINTER(_P)]),
    "gmupt_buffer_update": (C.c_int, [_P, _P, C.c_size_t]),
    "gmupt_buffer_destroy": (None, [_P]),
    "gmupt_buffer_size": (C.c_size_t, [_P]),
    "gmupt_texture_array_create": (C.c_int, [_P, _P, C.c_uint32, C.c_uint32, C.POINTER(_P)]),
    "gmupt_image_decode_png": (C.c_int, [_P, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(_P)]),
    "gmupt_image_free": (None, [_P]),
    "gmupt_image_resize_square": (C.c_int, [_P, C.c_uint32, C.c_uint32, _P]),
    "gmupt_texture_common_size": (C.c_uint32, [C.POINTER(C.c_size_t), C.c_uint32]),
    "gmupt_renderer_create": (C.c_int, [_P, C.POINTER(RendererDesc), C.POINTER(_P)]),
    "gmupt_renderer_destroy": (None, [_P]),
    "gmupt_renderer_bind_scene": (C.c_int, [_P] * 7),
    "gmupt_renderer_bind_textures": (C.c_int, [_P, _P, _P, _P]),
    "gmupt_set_camera": (C.c_int, [_P, C.POINTER(CameraBuffer)]),
    "gmupt_iterate": (C.c_int, [_P]),
    "gmupt_resize": (C.c_int, [_P, C.c_uint32, C.c_uint32]),
    "gmupt_read_framebuffer": (C.c_int, [_P, _P, C.c_size_t]),
    "gmupt_copy_framebuffer_to_device": (C.c_int, [_P, _P, C.c_size_t]),
    "gmupt_get_counters": (C.c_int, [_P, C.POINTER(C.c_uint32 * 8)]),
    "gmupt_synchronize": (C.c_int, [_P]),
    "gmupt_get_stats": (C.c_int, [_P, C.POINTER(Stats)]),
    "gmupt_reset_stats": (C.c_int, [_P]),
    "gmupt_enable_timing": (C.c_int, [_P, C.c_int]),
    "gmupt_render_budget": (C.c_int, [_P, _P, C.c_uint32, C.POINTER(C.c_uint32)]),
    "gmupt_debug_read_path_state": (C.c_int, [_P, _P, C.c_size_t]),
    "gmupt_debug_write_path_state": (C.c_int, [_P, _P, C.c_size_t]),
    "gmupt_debug_read_queues": (C.c_int, [_P, _P, C.c_size_t]),
    "gmupt_debug_write_queues": (C.c_int, [_P, _P, C.c_size_t]),
    "gmupt_debug_write_counters": (C.c_int, [_P, C.POINTER(C.c_uint32 * 8)]),
    "gmupt_debug_write_framebuffer": (C.c_int, [_P, _P, C.c_size_t]),
    "gmupt_debug_run_stage": (C.c_int, [_P, C.c_int]),
    "gmupt_debug_detmath": (C.c_int, [_P, C.c_int, _P, _P, _P, C.c_uint32]),
    "gmupt_sbvh_default_params": (None, [C.POINTER(SbvhParams)]),
    "gmupt_sbvh_build": (C.c_int, [_P, C.c_uint32, _P, C.c_uint32, C.POINTER(SbvhParams), C.POINTER(_P)]),
    "gmupt_sbvh_num_nodes": (C.c_uint32, [_P]),
    "gmupt_sbvh_num_references": (C.c_uint32, [_P]),
    "gmupt_sbvh_sah": (C.c_float, [_P]),
    "gmupt_sbvh_depth": (C.c_uint32, [_P]),
    "gmupt_sbvh_flatten": (C.c_int, [_P, _P, _P, _P, _P]),
    "gmupt_sbvh_destroy": (None, [_P]),
    "gmupt_camera_create": (C.c_int, [C.c_uint32, C.c_uint32, C.POINTER(_P)]),
    "gmupt_camera_destroy": (None, [_P]),
    "gmupt_camera_update_resolution": (None, [_P, C.c_uint32, C.c_uint32]),
    "gmupt_camera_set_pose": (None, [_P, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float]),
    "gmupt_camera_update": (None, [_P, C.c_float]),
    "gmupt_camera_set_input": (None, [_P, C.c_float, C.c_float, C.c_uint32]),
    "gmupt_camera_reset_accumulation": (None, [_P]),
    "gmupt_camera_get_buffer": (C.POINTER(CameraBuffer), [_P]),
    "gmupt_version": (C.c_char_p, []),
}

_lib = None
_libs = {}
_devices_created = 0   # gmupt_device_create calls of this process (a process that uses the GPU must not spawn a compiler)


def _load(path):
    if path not in _libs:
        handle = C.CDLL(path)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(handle, name)  # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _libs[path] = handle
    return _libs[path]


def lib():
    """Loads libgmupt.so (building it first if the sources are newer).  Fails loudly when it cannot."""
    global _lib
    if _lib is None:
        path = os.environ.get("GMUPT_LIB") or _build.LIB   # GMUPT_LIB: A/B timing of differently configured builds
        if not os.environ.get("GMUPT_LIB") and (not os.path.exists(path) or (os.path.exists("/opt/rocm/bin/hipcc") and _build.needs_build())):
            path = _build.build()
        _lib = _load(path)
    return _lib


class use_build:
    """Context manager: inside the block every call of this module goes to a test build of the same sources (build.TEST_BUILDS,
    e.g. "variants" = the library with the whole traversal ladder).  Objects created inside must be closed inside."""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        global _lib
        path = _build.lib_path(self.name)
        if not os.path.exists(path) or (os.path.exists("/opt/rocm/bin/hipcc") and _build.needs_build(self.name)):
            if _devices_created:
                # a process that has initialised the GPU must not start hipcc (fork + exec): the GPU box refuses it.  __graft_entry__.build()
                # prebuilds every test build; a stale one is an error here, not something to repair on the fly
                raise GmuptError("test build %r is missing or older than its sources and this process already uses the GPU: run __graft_entry__.build() first" % self.name)
            path = _build.build(name=self.name)
        self.saved = lib()
        _lib = _load(path)
        return _lib

    def __exit__(self, *exc):
        global _lib
        _lib = self.saved
        return False


def _check(rc):
    if rc != 0:
        raise GmuptError("gmupt error %d: %s" % (rc, lib().gmupt_last_error().decode(errors="replace")), rc)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Device:
    def __init__(self, index=0):
        global _devices_created
        self.h = _P()
        _check(lib().gmupt_device_create(index, C.byref(self.h)))
        _devices_created += 1

    def close(self):
        if self.h:
            lib().gmupt_device_destroy(self.h)
            self.h = _P()

    def detmath(self, fn, x, y=None):
        x = np.ascontiguousarray(x, dtype=np.float32)
        y = np.zeros_like(x) if y is None else np.ascontiguousarray(y, dtype=np.float32)
        out = np.empty_like(x)
        _check(lib().gmupt_debug_detmath(self.h, fn, _ptr(x), _ptr(y), _ptr(out), x.size))
        return out


class Buffer:
    def __init__(self, dev, kind, array):
        array = np.ascontiguousarray(array)
        self.h = _P()
        self.dev = dev
        _check(lib().gmupt_buffer_create(dev.h, kind, _ptr(array), array.nbytes, C.byref(self.h)))

    def update(self, array):
        array = np.ascontiguousarray(array)
        _check(lib().gmupt_buffer_update(self.h, _ptr(array), array.nbytes))

    def close(self):
        if self.h:
            lib().gmupt_buffer_destroy(self.h)
            self.h = _P()


class TextureArray:
    """R8G8B8A8_UNORM Texture2DArray: (layers, size, size, 4) uint8."""

    def __init__(self, dev, array):
        array = np.ascontiguousarray(array, dtype=np.uint8)
        assert array.ndim == 4 and array.shape[1] == array.shape[2] and array.shape[3] == 4
        self.h = _P()
        _check(lib().gmupt_texture_array_create(dev.h, _ptr(array), array.shape[1], array.shape[0], C.byref(self.h)))

    def close(self):
        if self.h:
            lib().gmupt_buffer_destroy(self.h)
            self.h = _P()


def decode_png(data):
    """PNG file bytes -> (h, w, 4) uint8 through gmupt_image_decode_png (the reference's lodepng::decode call, Scene.cpp:226)."""
    w, h, mem = C.c_uint32(), C.c_uint32(), _P()
    buf = (C.c_uint8 * len(data)).from_buffer_copy(bytes(data))
    _check(lib().gmupt_image_decode_png(buf, len(data), C.byref(w), C.byref(h), C.byref(mem)))
    try:
        out = np.ctypeslib.as_array(C.cast(mem, C.POINTER(C.c_uint8)), shape=(h.value, w.value, 4)).copy()
    finally:
        lib().gmupt_image_free(mem)
    return out


def resize_square(rgba, new_size):
    """Square RGBA8 resize: the bytes of the reference's avir call (Scene.cpp:276-279; host/AvirResize.cpp restates avir's pipeline for it)."""
    rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
    assert rgba.ndim == 3 and rgba.shape[0] == rgba.shape[1] and rgba.shape[2] == 4
    out = np.empty((new_size, new_size, 4), dtype=np.uint8)
    _check(lib().gmupt_image_resize_square(_ptr(rgba), rgba.shape[0], new_size, _ptr(out)))
    return out


def texture_common_size(layer_bytes):
    arr = (C.c_size_t * len(layer_bytes))(*layer_bytes)
    return int(lib().gmupt_texture_common_size(arr, len(layer_bytes)))


def texture_array_from_png(files):
    """Layers from encoded PNG files in material order, as Scene::loadSpecificTexture + createTextures: decode, pick the common size by
    the median rule, resize the layers that differ.  Returns (layers, size, size, 4) uint8."""
    layers = [decode_png(f) for f in files]
    for l in layers:
        if l.shape[0] != l.shape[1]:
            raise GmuptError("texture is not square")
    size = texture_common_size([l.size for l in layers])
    return np.stack([l if l.shape[0] == size else resize_square(l, size) for l in layers])


class SceneBuffers:
    """The six scene resources of Renderer::draw (t0-t4, b1) uploaded through gmupt_buffer_create."""

    def __init__(self, dev, scene):
        self.nodes = Buffer(dev, BUFFER_BVH_NODES, scene["nodes"])
        self.tris = Buffer(dev, BUFFER_TRIANGLES, scene["tris"])
        self.verts = Buffer(dev, BUFFER_VERTICES, scene["verts"])
        self.lights = Buffer(dev, BUFFER_LIGHTS, scene["lights"])
        self.props = Buffer(dev, BUFFER_TRI_PROPS, scene["props"])
        self.materials = Buffer(dev, BUFFER_MATERIALS, scene["materials"])
        self.textures = [TextureArray(dev, scene[k]) if scene.get(k) is not None else None
                         for k in ("tex_diffuse", "tex_metallic_roughness", "tex_normal")]

    def all(self):
        return [self.nodes, self.tris, self.verts, self.lights, self.props, self.materials]

    def close(self):
        for b in self.all():
            b.close()
        for t in self.textures:
            if t is not None:
                t.close()


class Camera:
    """Host camera (gmupt_camera_*): Camera::update / setRotation / updateResolution of the reference."""

    def __init__(self, width, height):
        self.h = _P()
        _check(lib().gmupt_camera_create(width, height, C.byref(self.h)))

    def set_pose(self, x, y, z, pitch, yaw):
        lib().gmupt_camera_set_pose(self.h, x, y, z, pitch, yaw)

    def update(self, dt=0.0):
        lib().gmupt_camera_update(self.h, dt)

    def update_resolution(self, width, height):
        lib().gmupt_camera_update_resolution(self.h, width, height)

    def reset_accumulation(self):
        lib().gmupt_camera_reset_accumulation(self.h)

    def set_input(self, mouse_dx=0.0, mouse_dy=0.0, w=False, s=False, a=False, d=False):
        lib().gmupt_camera_set_input(self.h, mouse_dx, mouse_dy, (1 if w else 0) | (2 if s else 0) | (4 if a else 0) | (8 if d else 0))

    @property
    def buffer(self):
        return lib().gmupt_camera_get_buffer(self.h).contents

    def buffer_copy(self):
        out = CameraBuffer()
        C.memmove(C.byref(out), C.byref(self.buffer), C.sizeof(CameraBuffer))
        return out

    def close(self):
        if self.h:
            lib().gmupt_camera_destroy(self.h)
            self.h = _P()


class Renderer:
    def __init__(self, dev, width, height, pool_paths=0, live_paths=0, tile=None, path_budget=0, max_depth=0, collect_stats=False):
        d = RendererDesc(width, height, pool_paths, live_paths, 0, 0, 0, path_budget, max_depth, 1 if collect_stats else 0)
        if tile is not None:
            d.tile_enabled, d.tile_x0, d.tile_y0 = 1, tile[0], tile[1]
        self.desc = d
        self.dev = dev
        self.h = _P()
        _check(lib().gmupt_renderer_create(dev.h, C.byref(d), C.byref(self.h)))
        self.width, self.height = width, height
        self.pool = pool_paths or PATHCOUNT
        self._scene = None

    def bind_scene(self, sb):
        self._scene = sb  # keep the buffers alive
        _check(lib().gmupt_renderer_bind_scene(self.h, *[b.h for b in sb.all()]))
        _check(lib().gmupt_renderer_bind_textures(self.h, *[(t.h if t is not None else None) for t in sb.textures]))

    def set_camera(self, cam_buffer):
        _check(lib().gmupt_set_camera(self.h, C.byref(cam_buffer)))

    def iterate(self):
        _check(lib().gmupt_iterate(self.h))

    def run_stage(self, stage):
        _check(lib().gmupt_debug_run_stage(self.h, stage))

    def synchronize(self):
        _check(lib().gmupt_synchronize(self.h))

    def resize(self, w, h):
        _check(lib().gmupt_resize(self.h, w, h))
        self.width, self.height = w, h

    def framebuffer(self):
        out = np.empty((self.height, self.width, 4), dtype=np.float32)
        _check(lib().gmupt_read_framebuffer(self.h, _ptr(out), out.nbytes))
        return out

    def copy_framebuffer_to_device(self, device_ptr, nbytes):
        _check(lib().gmupt_copy_framebuffer_to_device(self.h, C.c_void_p(device_ptr), nbytes))

    def counters(self):
        out = (C.c_uint32 * 8)()
        _check(lib().gmupt_get_counters(self.h, C.byref(out)))
        return np.array(list(out), dtype=np.uint32)

    def stats(self, check=True):
        """gmupt_get_stats.  A launch that flagged its results as invalid makes the call fail (GMUPT_ERR_CAST_FAULT) although the
        statistics are filled in: check=False returns them anyway (to look at the flags)."""
        s = Stats()
        rc = lib().gmupt_get_stats(self.h, C.byref(s))
        if rc != ERR_CAST_FAULT or check:
            _check(rc)
        return s

    def reset_stats(self):
        _check(lib().gmupt_reset_stats(self.h))

    def enable_timing(self, mode=1):
        _check(lib().gmupt_enable_timing(self.h, int(mode)))

    def render_budget(self, camera, max_iterations=1 << 20):
        it = C.c_uint32(0)
        _check(lib().gmupt_render_budget(self.h, camera.h, max_iterations, C.byref(it)))
        return it.value

    # reference-layout debug access
    def read_path_state(self):
        out = np.empty(self.pool * STATE_BYTES, dtype=np.uint8)
        _check(lib().gmupt_debug_read_path_state(self.h, _ptr(out), out.nbytes))
        return out

    def write_path_state(self, raw):
        raw = np.ascontiguousarray(raw, dtype=np.uint8)
        _check(lib().gmupt_debug_write_path_state(self.h, _ptr(raw), raw.nbytes))

    def read_queues(self):
        out = np.empty(self.pool * 5, dtype=np.uint32)
        _check(lib().gmupt_debug_read_queues(self.h, _ptr(out), out.nbytes))
        return out.reshape(5, self.pool)

    def write_queues(self, q):
        q = np.ascontiguousarray(q, dtype=np.uint32)
        _check(lib().gmupt_debug_write_queues(self.h, _ptr(q), q.nbytes))

    def write_counters(self, qc):
        arr = (C.c_uint32 * 8)(*[int(v) for v in qc])
        _check(lib().gmupt_debug_write_counters(self.h, C.byref(arr)))

    def write_framebuffer(self, fb):
        fb = np.ascontiguousarray(fb, dtype=np.float32)
        _check(lib().gmupt_debug_write_framebuffer(self.h, _ptr(fb), fb.nbytes))

    def close(self):
        if self.h:
            lib().gmupt_renderer_destroy(self.h)
            self.h = _P()


def sbvh_build(verts, indices, vertex_material=None, params=None):
    """Host SBVH build + flatten (gmupt_sbvh_*).  Returns dict(nodes, tris, ref_triangle, sah, depth)."""
    verts = np.ascontiguousarray(verts, dtype=np.float32).reshape(-1, 3)
    indices = np.ascontiguousarray(indices, dtype=np.int32).reshape(-1, 3)
    h = _P()
    pp = None
    if params is not None:
        pp = SbvhParams()
        lib().gmupt_sbvh_default_params(C.byref(pp))
        for k, v in params.items():
            setattr(pp, k, v)
        pp = C.byref(pp)
    _check(lib().gmupt_sbvh_build(_ptr(verts), verts.shape[0], _ptr(indices), indices.shape[0], pp, C.byref(h)))
    try:
        n = lib().gmupt_sbvh_num_nodes(h)
        nr = lib().gmupt_sbvh_num_references(h)
        nodes = np.zeros(n, dtype=bvh_node_dtype)
        tris = np.zeros(max(nr, 1), dtype=triangle_dtype)[:nr]
        ref = np.zeros(max(nr, 1), dtype=np.int32)[:nr]
        vm = None
        if vertex_material is not None:
            vm = np.ascontiguousarray(vertex_material, dtype=np.uint32)
        tris_buf = np.zeros(max(nr, 1), dtype=triangle_dtype)
        ref_buf = np.zeros(max(nr, 1), dtype=np.int32)
        _check(lib().gmupt_sbvh_flatten(h, _ptr(vm) if vm is not None else None, _ptr(nodes), _ptr(tris_buf), _ptr(ref_buf)))
        tris, ref = tris_buf[:nr], ref_buf[:nr]
        return {"nodes": nodes, "tris": tris, "ref_triangle": ref, "sah": float(lib().gmupt_sbvh_sah(h)), "depth": int(lib().gmupt_sbvh_depth(h))}
    finally:
        lib().gmupt_sbvh_destroy(h)
