"""Generates tests/golden/texture_ref.npz with the REFERENCE's own vendored texture libraries (build container only).

oracle/_ref/libreftex.so = lodepng + avir compiled from /root/reference/Include by oracle/Makefile.  For a few small synthetic layers the file
holds: the PNG bytes, what lodepng::decode makes of them (Scene.cpp:226), and what avir's CImageResizer<fpclass_float8_dil>(8) makes of the
square RGBA8 layer at the sizes the median rule can ask for (Scene.cpp:269-279) -- the pins of gmupt_image_decode_png / gmupt_image_resize_square.
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gmupt_pkg
import png_util


def ref_lib():
    L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libreftex.so"))
    L.ref_lodepng_decode.restype = C.c_uint
    L.ref_lodepng_decode.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.POINTER(C.c_void_p)]
    L.ref_free.argtypes = [C.c_void_p]
    L.ref_avir_resize_square.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_uint]
    return L


def ref_decode(L, data):
    w, h, mem = C.c_uint(), C.c_uint(), C.c_void_p()
    err = L.ref_lodepng_decode(data, len(data), C.byref(w), C.byref(h), C.byref(mem))
    if err:
        raise RuntimeError("lodepng error %d" % err)
    out = np.ctypeslib.as_array(C.cast(mem, C.POINTER(C.c_uint8)), shape=(h.value, w.value, 4)).copy()
    L.ref_free(mem)
    return out


def ref_resize(L, rgba, new):
    rgba = np.ascontiguousarray(rgba, np.uint8)
    out = np.empty((new, new, 4), np.uint8)
    L.ref_avir_resize_square(rgba.ctypes.data, rgba.shape[0], out.ctypes.data, new)
    return out


def layers():
    """name -> square RGBA8 image: smooth, periodic, sharp-edged and noisy content, power-of-two and odd sizes."""
    rng = np.random.default_rng(11)
    out = {}
    for size in (16, 24, 64):
        yy, xx = np.mgrid[0:size, 0:size].astype(np.float64)
        img = np.zeros((size, size, 4), np.uint8)
        img[..., 0] = np.round(255 * xx / (size - 1)); img[..., 1] = np.round(255 * yy / (size - 1))
        img[..., 2] = np.round(127.5 + 127.5 * np.sin(xx * 0.7) * np.cos(yy * 0.4)); img[..., 3] = 255
        out["smooth%d" % size] = img
        chk = np.zeros((size, size, 4), np.uint8)
        chk[..., :3] = np.where(((xx // 4 + yy // 4) % 2)[..., None] == 0, (235, 235, 235), (20, 40, 200)); chk[..., 3] = np.where(xx < size // 2, 255, 128)
        out["checker%d" % size] = chk
    out["noise32"] = rng.integers(0, 256, (32, 32, 4)).astype(np.uint8)
    # layers of the second group (EXTRA_TARGETS): sizes at which avir picks its other filter structures
    rng2 = np.random.default_rng(23)
    for size in (1, 7, 37, 63):
        out["grain%d" % size] = rng2.integers(0, 256, (size, size, 4)).astype(np.uint8)
    yy, xx = np.mgrid[0:200, 0:200].astype(np.float64)
    img = np.zeros((200, 200, 4), np.uint8)
    img[..., 0] = np.round(127.5 + 127.5 * np.sin(xx * 0.9 + yy * 0.13)); img[..., 1] = np.round(255 * ((xx * yy) % 4096) / 4095)
    img[..., 2] = np.where((xx.astype(int) ^ yy.astype(int)) & 8, 250, 5); img[..., 3] = np.round(255 * yy / 199)
    out["waves200"] = img
    return out


# second group: (layer, new size) pairs chosen so that the fixture holds every filter structure avir's cost model picks for this call -- both
# interpolation orders, filter and interpolator separate and combined, a different structure for rows and columns (smooth64 -> 300), the
# 16-fold half-band step of reductions beyond 32:1 (waves200 -> 6), equal sizes, one-texel images
EXTRA_TARGETS = (("grain1", 7), ("grain7", 1), ("grain7", 64), ("grain37", 37), ("grain37", 100), ("grain37", 17), ("grain63", 64), ("grain63", 62),
                 ("waves200", 128), ("waves200", 33), ("waves200", 6), ("waves200", 201), ("smooth64", 300), ("noise32", 32))


def targets(name, size):
    if name.startswith(("grain", "waves")):
        return []
    return sorted({16, 24, 32, 48, 64, 128} - {size})


def main():
    L = ref_lib()
    data = {}
    for name, img in layers().items():
        png = png_util.encode(img, 6, 8, filters=(0, 1, 2, 3, 4), level=6)
        data["png_" + name] = np.frombuffer(png, np.uint8)
        data["decoded_" + name] = ref_decode(L, png)
        assert np.array_equal(data["decoded_" + name], img)
        size = img.shape[0]
        for new in targets(name, size):
            data["resized_%s_to%d" % (name, new)] = ref_resize(L, img, new)
    for name, new in EXTRA_TARGETS:
        data["resized_%s_to%d" % (name, new)] = ref_resize(L, layers()[name], new)
    # other PNG flavours through lodepng (grey, palette + tRNS, 16 bit, interlaced): decode pins only
    rng = np.random.default_rng(5)
    for ct, depth in ((0, 8), (0, 16), (2, 16), (3, 4), (4, 8)):
        ch = png_util.CHANNELS[ct]
        samples = rng.integers(0, 1 << depth, (9, 14, ch))
        palette = rng.integers(0, 256, ((1 << depth), 3)) if ct == 3 else None
        trns = bytes(rng.integers(0, 256, 5).astype(np.uint8)) if ct == 3 else None
        png = png_util.encode(samples, ct, depth, filters=(0, 2, 4), interlace=(ct == 3), palette=palette, trns=trns)
        data["png_fmt%d_%d" % (ct, depth)] = np.frombuffer(png, np.uint8)
        data["decoded_fmt%d_%d" % (ct, depth)] = ref_decode(L, png)
    path = os.path.join(ROOT, "tests", "golden", "texture_ref.npz")
    np.savez_compressed(path, **data)
    print("wrote %s: %d arrays, %d bytes" % (path, len(data), os.path.getsize(path)))
    # the product's resize against avir (a restatement of avir's pipeline: every line must say 0)
    capi = gmupt_pkg.load().capi
    for k in sorted(data):
        if k.startswith("resized_"):
            name, new = k[len("resized_"):].rsplit("_to", 1)
            mine = capi.resize_square(layers()[name], int(new)).astype(int)
            d = np.abs(mine - data[k].astype(int))
            print("%-28s max |diff| %3d  mean %.3f  > 1: %.2f %%" % (k, d.max(), d.mean(), 100.0 * (d > 1).mean()))


if __name__ == "__main__":
    main()
