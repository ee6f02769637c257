"""f3, host side: PNG decoding, the common-size rule and the square resize that feed the texture arrays
(reference Source/Scene.cpp:209-290: lodepng::decode, median of the distinct sizes, avir resize).  No GPU."""
import os
import struct
import zlib

import numpy as np
import pytest

import png_util


@pytest.mark.parametrize("color_type,depth", [(0, 1), (0, 2), (0, 4), (0, 8), (0, 16), (2, 8), (2, 16), (3, 1), (3, 2), (3, 4), (3, 8), (4, 8), (4, 16), (6, 8), (6, 16)])
@pytest.mark.parametrize("interlace", [False, True])
def test_png_decoder_all_formats(pkg, color_type, depth, interlace):
    rng = np.random.default_rng(color_type * 100 + depth + (7 if interlace else 0))
    h, w = 13, 21                                  # odd sizes: partial bytes at row ends, empty / ragged Adam7 passes
    ch = png_util.CHANNELS[color_type]
    palette = rng.integers(0, 256, ((1 << depth), 3)) if color_type == 3 else None
    samples = rng.integers(0, 1 << depth, (h, w, ch))
    trns = None
    if color_type == 3:
        trns = bytes(rng.integers(0, 256, max(1, (1 << depth) // 2)).astype(np.uint8))
    elif color_type == 0:
        trns = struct.pack(">H", int(samples[3, 4, 0]))
    elif color_type == 2:
        trns = struct.pack(">HHH", *[int(v) for v in samples[5, 6]])
    data = png_util.encode(samples, color_type, depth, filters=(0, 1, 2, 3, 4), interlace=interlace, palette=palette, trns=trns, idat_split=37)
    got = pkg.capi.decode_png(data)
    assert got.shape == (h, w, 4)
    assert np.array_equal(got, png_util.expected_rgba(samples, color_type, depth, palette, trns))


def test_png_decoder_agrees_with_pillow_and_handles_deflate_block_types(pkg, tmp_path):
    from PIL import Image
    rng = np.random.default_rng(3)
    img = np.zeros((64, 64, 4), np.uint8)
    yy, xx = np.mgrid[0:64, 0:64]
    img[..., 0] = 4 * xx; img[..., 1] = 4 * yy; img[..., 2] = ((xx // 8 + yy // 8) % 2) * 255; img[..., 3] = 255
    img[20:40, 20:40] = rng.integers(0, 256, (20, 20, 4))
    for level in (0, 1, 9):                        # stored blocks, fixed/dynamic Huffman with long matches
        data = png_util.encode(img, 6, 8, filters=(4,), level=level)
        assert np.array_equal(pkg.capi.decode_png(data), img)
    path = tmp_path / "pil.png"
    Image.fromarray(img).save(str(path), optimize=True)
    assert np.array_equal(pkg.capi.decode_png(path.read_bytes()), img)
    Image.fromarray(img[..., :3]).convert("P", palette=Image.ADAPTIVE, colors=16).save(str(path))
    assert np.array_equal(pkg.capi.decode_png(path.read_bytes()), np.asarray(Image.open(str(path)).convert("RGBA")))
    # a 1x1 image and a large flat one (long runs: overlapping copies)
    assert np.array_equal(pkg.capi.decode_png(png_util.encode(img[:1, :1], 6, 8)), img[:1, :1])
    flat = np.full((300, 300, 4), 77, np.uint8)
    assert np.array_equal(pkg.capi.decode_png(png_util.encode(flat, 6, 8, level=9)), flat)


def test_png_decoder_rejects_damaged_files(pkg):
    img = np.random.default_rng(1).integers(0, 256, (8, 8, 4)).astype(np.uint8)
    good = png_util.encode(img, 6, 8)
    E = pkg.capi.GmuptError
    with pytest.raises(E, match="signature"):
        pkg.capi.decode_png(b"JUNK" + good[4:])
    bad = bytearray(good); bad[40] ^= 0x55
    with pytest.raises(E, match="CRC"):
        pkg.capi.decode_png(bytes(bad))
    with pytest.raises(E):
        pkg.capi.decode_png(good[:len(good) // 2])
    # valid container, corrupt compressed stream (Adler-32)
    raw = b"".join(b"\x00" + img[y].tobytes() for y in range(8))
    z = bytearray(zlib.compress(raw)); z[-1] ^= 1
    broken = good[:8] + png_util.chunk(b"IHDR", struct.pack(">IIBBBBB", 8, 8, 8, 6, 0, 0, 0)) + png_util.chunk(b"IDAT", bytes(z)) + png_util.chunk(b"IEND", b"")
    with pytest.raises(E, match="Adler"):
        pkg.capi.decode_png(broken)
    with pytest.raises(E, match="bit depth"):
        pkg.capi.decode_png(good[:8] + png_util.chunk(b"IHDR", struct.pack(">IIBBBBB", 8, 8, 4, 6, 0, 0, 0)) + good[33:])
    with pytest.raises(E, match="critical"):
        pkg.capi.decode_png(good[:33] + png_util.chunk(b"XXXX", b"1") + good[33:])


def test_common_size_is_the_median_of_the_distinct_sizes(pkg):
    f = pkg.capi.texture_common_size
    b = lambda n: n * n * 4
    assert f([b(256)]) == 256
    assert f([b(256), b(512)]) == 512                         # two distinct sizes: element [1]
    assert f([b(256), b(256), b(256), b(1024)]) == 1024       # duplicates do not count (std::set, Scene.cpp:212,228)
    assert f([b(1024), b(64), b(512)]) == 512
    assert f([b(2048), b(64), b(512), b(128)]) == 512
    assert f([]) == 0


def test_resize_square_properties(pkg):
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (32, 32, 4)).astype(np.uint8)
    assert np.array_equal(pkg.capi.resize_square(img, 32), img)               # same size: passed through bit for bit
    flat = np.full((48, 48, 4), (10, 128, 255, 77), np.uint8)
    for n in (16, 31, 48, 100):
        assert np.array_equal(pkg.capi.resize_square(flat, n), np.full((n, n, 4), (10, 128, 255, 77), np.uint8))   # weights sum to 1
    yy, xx = np.mgrid[0:64, 0:64]
    ramp = np.stack([3 * xx + 10, 3 * yy + 20, xx + yy + 30, np.full_like(xx, 255)], axis=-1).astype(np.uint8)
    half = pkg.capi.resize_square(ramp, 32)
    box = ramp.reshape(32, 2, 32, 2, 4).mean(axis=(1, 3))
    assert np.abs(half[2:-2, 2:-2].astype(np.float64) - box[2:-2, 2:-2]).max() <= 1.0     # a linear ramp is reproduced away from the clamped edge
    up = pkg.capi.resize_square(ramp, 128)
    ref = np.broadcast_to(3 * ((np.arange(128) + 0.5) / 2 - 0.5) + 10, (128, 128))      # red = 3 x + 10 at the source position of each texel
    assert np.abs(up[8:-8, 8:-8, 0].astype(np.float64) - ref[8:-8, 8:-8]).max() <= 1.0
    # a checkerboard at the Nyquist rate shrinks to its mean: the filter is stretched when reducing (no aliasing)
    chk = np.where(((xx + yy) % 2)[..., None] == 0, 255, 0).astype(np.uint8).repeat(4, axis=-1)
    small = pkg.capi.resize_square(chk, 16)
    assert np.abs(small[2:-2, 2:-2].astype(np.float64) - 127.5).max() <= 8.0
    with pytest.raises(pkg.capi.GmuptError):
        pkg.capi._check(pkg.capi.lib().gmupt_image_resize_square(None, 4, 4, None))


def test_texture_array_from_png_resizes_to_the_common_size(pkg):
    rng = np.random.default_rng(2)
    a = rng.integers(0, 256, (16, 16, 4)).astype(np.uint8)
    b = rng.integers(0, 256, (32, 32, 4)).astype(np.uint8)
    c = rng.integers(0, 256, (32, 32, 4)).astype(np.uint8)
    files = [pkg.scenes.encode_png_rgba8(x) for x in (a, b, c)]
    arr = pkg.capi.texture_array_from_png(files)
    assert arr.shape == (3, 32, 32, 4)
    assert np.array_equal(arr[1], b) and np.array_equal(arr[2], c) and np.array_equal(arr[0], pkg.capi.resize_square(a, 32))
    with pytest.raises(pkg.capi.GmuptError, match="square"):
        pkg.capi.texture_array_from_png([pkg.scenes.encode_png_rgba8(np.zeros((4, 8, 4), np.uint8))])


def test_png_decoder_random_images_property(pkg):
    # property test: any image our spec-level encoder can write (random size, colour type, depth, filters per row, interlace, IDAT
    # split, compression level) decodes to the RGBA8 image the conversion rules give
    from hypothesis import given, settings, strategies as st

    formats = [(0, 1), (0, 2), (0, 4), (0, 8), (0, 16), (2, 8), (2, 16), (3, 1), (3, 2), (3, 4), (3, 8), (4, 8), (4, 16), (6, 8), (6, 16)]

    @settings(max_examples=120, deadline=None)
    @given(st.integers(0, len(formats) - 1), st.integers(1, 19), st.integers(1, 19), st.booleans(), st.integers(0, 2 ** 31 - 1),
           st.lists(st.integers(0, 4), min_size=1, max_size=5), st.sampled_from([0, 1, 6, 9]), st.sampled_from([0, 1, 7, 64]))
    def check(fmt, w, h, interlace, seed, filters, level, split):
        color_type, depth = formats[fmt]
        rng = np.random.default_rng(seed)
        ch = png_util.CHANNELS[color_type]
        samples = rng.integers(0, 1 << depth, (h, w, ch))
        palette = rng.integers(0, 256, ((1 << depth), 3)) if color_type == 3 else None
        trns = bytes(rng.integers(0, 256, 1 << depth).astype(np.uint8))[: int(rng.integers(0, (1 << depth) + 1))] if color_type == 3 else None
        if trns is not None and len(trns) == 0:
            trns = None
        data = png_util.encode(samples, color_type, depth, filters=tuple(filters), interlace=interlace, palette=palette, trns=trns, level=level, idat_split=split)
        got = pkg.capi.decode_png(data)
        assert np.array_equal(got, png_util.expected_rgba(samples, color_type, depth, palette, trns))

    check()


# ---- pins against the reference's own vendored libraries (fixture: tests/golden/texture_ref.npz, made by tools/make_texture_golden.py from
# oracle/_ref/libreftex.so = lodepng + avir compiled from /root/reference/Include in the build container)
def _texture_fixture():
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "texture_ref.npz"))


def test_png_decoder_matches_lodepng_fixture(pkg):
    # gmupt_image_decode_png replaces lodepng::decode (Source/Scene.cpp:226): same RGBA8 bytes for every fixture file
    z = _texture_fixture()
    names = [k[4:] for k in z.files if k.startswith("png_")]
    assert len(names) >= 12
    for name in names:
        got = pkg.capi.decode_png(z["png_" + name].tobytes())
        assert np.array_equal(got, z["decoded_" + name]), name


def test_resize_against_avir_fixture(pkg):
    # gmupt_image_resize_square replaces avir::CImageResizer<fpclass_float8_dil>(8)::resizeImage (Source/Scene.cpp:269-279) with a restatement of
    # avir's pipeline for that call (host/AvirResize.cpp): every byte of every fixture case, bit for bit.  The fixture's second group
    # (tools/make_texture_golden.py EXTRA_TARGETS) holds the sizes at which avir's cost model picks its other filter structures: both
    # interpolation orders, filter and interpolator combined and separate, different structures for rows and columns, the half-band step of
    # reductions beyond 32:1, equal sizes (avir runs its filters then, too; the fixture's two cases come back as the source) and one-texel images.
    import make_texture_golden as M
    z = _texture_fixture()
    layers = M.layers()
    keys = [k for k in z.files if k.startswith("resized_")]
    assert len(keys) >= 45
    for name, new in M.EXTRA_TARGETS:
        assert "resized_%s_to%d" % (name, new) in keys
    for k in keys:
        name, new = k[len("resized_"):].rsplit("_to", 1)
        new = int(new); src = layers[name]
        mine = pkg.capi.resize_square(src, new)
        assert mine.shape == z[k].shape and np.array_equal(mine, z[k]), (k, int(np.abs(mine.astype(int) - z[k].astype(int)).max()))


def test_resize_against_avir_live():
    # build container only: random sizes and contents against avir compiled from the reference tree (oracle/_ref/libreftex.so), bit for bit
    ref = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libreftex.so")
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref/libreftex.so is only built where /root/reference exists")
    import gmupt_pkg
    import make_texture_golden as M
    capi = gmupt_pkg.load().capi
    L = M.ref_lib()
    rng = np.random.default_rng(77)
    pairs = [(int(rng.integers(1, 260)), int(rng.integers(1, 260))) for _ in range(40)] + [(512, 96), (96, 512), (300, 9), (2, 2), (1, 1), (256, 255), (255, 256)]
    for old, new in pairs:
        src = rng.integers(0, 256, (old, old, 4)).astype(np.uint8)
        if (old + new) & 1:                                      # half of the cases: smooth content (long runs of equal rounding decisions)
            yy, xx = np.mgrid[0:old, 0:old]
            src[..., 0] = (xx * 255) // max(old - 1, 1); src[..., 1] = (yy * 255) // max(old - 1, 1); src[..., 3] = 255
        assert np.array_equal(capi.resize_square(src, new), M.ref_resize(L, src, new)), (old, new)


def test_texture_fixture_reproduces_from_the_reference_libraries():
    # where the reference tree is present (the build container) the fixture is re-derived from lodepng / avir and must match bit for bit
    ref = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libreftex.so")
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref/libreftex.so is only built where /root/reference exists")
    import make_texture_golden as M
    L = M.ref_lib()
    z = _texture_fixture()
    layers = M.layers()
    for k in z.files:
        if k.startswith("decoded_"):
            assert np.array_equal(M.ref_decode(L, z["png_" + k[8:]].tobytes()), z[k]), k
        elif k.startswith("resized_"):
            name, new = k[len("resized_"):].rsplit("_to", 1)
            assert np.array_equal(M.ref_resize(L, layers[name], int(new)), z[k]), k
