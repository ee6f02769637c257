"""PNG encoder for the decoder tests: every colour type / bit depth / filter / Adam7, written from the PNG specification."""
import struct
import zlib

import numpy as np

CHANNELS = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}


def _paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)


def _pack_rows(samples, depth):
    """samples: (h, w, channels) integer array of `depth`-bit values -> list of packed row byte strings."""
    h, w, ch = samples.shape
    rows = []
    for y in range(h):
        flat = samples[y].reshape(-1).astype(np.uint32)
        if depth == 8:
            rows.append(flat.astype(np.uint8).tobytes())
        elif depth == 16:
            rows.append(flat.astype(">u2").tobytes())
        else:
            bits = np.zeros(((flat.size * depth + 7) // 8) * 8, np.uint8)
            for k in range(depth):
                bits[k::depth][:flat.size] = (flat >> (depth - 1 - k)) & 1
            rows.append(np.packbits(bits).tobytes())
    return rows


def _filter_rows(rows, bpp, filters):
    out, prev = bytearray(), None
    for y, row in enumerate(rows):
        f = filters[y % len(filters)]
        cur = bytearray(row)
        enc = bytearray(len(cur))
        for i in range(len(cur)):
            a = cur[i - bpp] if i >= bpp else 0
            b = prev[i] if prev is not None else 0
            c = prev[i - bpp] if (prev is not None and i >= bpp) else 0
            pred = (0, a, b, (a + b) >> 1, _paeth(a, b, c))[f]
            enc[i] = (cur[i] - pred) & 255
        out.append(f); out.extend(enc)
        prev = cur
    return bytes(out)


def chunk(kind, body):
    return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body) & 0xFFFFFFFF)


def encode(samples, color_type, depth, filters=(0,), interlace=False, palette=None, trns=None, level=6, idat_split=0):
    """samples: (h, w, channels) array of raw sample values at `depth` bits (palette indices for colour type 3)."""
    samples = np.asarray(samples)
    h, w, ch = samples.shape
    assert ch == CHANNELS[color_type]
    bits = ch * depth
    bpp = max(1, bits // 8)
    if not interlace:
        raw = _filter_rows(_pack_rows(samples, depth), bpp, filters)
    else:
        x0 = (0, 4, 0, 2, 0, 1, 0); y0 = (0, 0, 4, 0, 2, 0, 1); dx = (8, 8, 4, 4, 2, 2, 1); dy = (8, 8, 8, 4, 4, 2, 2)
        raw = b""
        for p in range(7):
            sub = samples[y0[p]::dy[p], x0[p]::dx[p]]
            if sub.shape[0] == 0 or sub.shape[1] == 0:
                continue
            raw += _filter_rows(_pack_rows(sub, depth), bpp, filters)
    data = zlib.compress(raw, level)
    out = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, color_type, 0, 0, 1 if interlace else 0))
    if palette is not None:
        out += chunk(b"PLTE", np.asarray(palette, np.uint8).tobytes())
    if trns is not None:
        out += chunk(b"tRNS", bytes(trns))
    out += chunk(b"tEXt", b"Comment\x00ancillary chunks are skipped")
    if idat_split:
        for i in range(0, len(data), idat_split):
            out += chunk(b"IDAT", data[i:i + idat_split])
    else:
        out += chunk(b"IDAT", data)
    return out + chunk(b"IEND", b"")


def expected_rgba(samples, color_type, depth, palette=None, trns=None):
    """The RGBA8 image a decoder following the LCT_RGBA / 8-bit conversion rules returns."""
    samples = np.asarray(samples).astype(np.uint32)
    h, w, _ = samples.shape
    maxv = (1 << depth) - 1

    def to8(v):
        return (v if depth == 8 else (v >> 8) if depth == 16 else (v * 255) // maxv).astype(np.uint8)
    out = np.zeros((h, w, 4), np.uint8)
    if color_type == 0:
        out[..., 0] = out[..., 1] = out[..., 2] = to8(samples[..., 0]); out[..., 3] = 255
        if trns is not None:
            key = (trns[0] << 8) | trns[1]
            out[..., 3] = np.where(samples[..., 0] == key, 0, 255)
    elif color_type == 2:
        out[..., :3] = to8(samples); out[..., 3] = 255
        if trns is not None:
            key = [(trns[2 * c] << 8) | trns[2 * c + 1] for c in range(3)]
            out[..., 3] = np.where((samples[..., 0] == key[0]) & (samples[..., 1] == key[1]) & (samples[..., 2] == key[2]), 0, 255)
    elif color_type == 3:
        pal = np.asarray(palette, np.uint8).reshape(-1, 3)
        out[..., :3] = pal[samples[..., 0]]
        alpha = np.full(pal.shape[0], 255, np.uint8)
        if trns is not None:
            alpha[:len(trns)] = np.frombuffer(bytes(trns), np.uint8)
        out[..., 3] = alpha[samples[..., 0]]
    elif color_type == 4:
        out[..., 0] = out[..., 1] = out[..., 2] = to8(samples[..., 0]); out[..., 3] = to8(samples[..., 1])
    else:
        out[...] = to8(samples)
    return out
