"""Tiny PNG writer for tests (8-bit grayscale, filter type 0 or Paeth per row)."""
import struct
import zlib

import numpy as np


def _chunk(t, d):
    return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)


def write_gray_png(path, img, paeth=True):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    flat = img.astype(np.int32)
    raw = bytearray()
    for y in range(h):
        cur = flat[y]
        if paeth and y % 2:
            up = flat[y - 1]
            a = np.concatenate([[0], cur[:-1]]); c = np.concatenate([[0], up[:-1]])
            pp = a + up - c
            pa, pb, pc = np.abs(pp - a), np.abs(pp - up), np.abs(pp - c)
            pred = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, up, c))
            raw += b"\x04" + ((cur - pred) & 255).astype(np.uint8).tobytes()
        else:
            raw += b"\x00" + img[y].tobytes()
    z = zlib.compress(bytes(raw), 6)
    with open(path, "wb") as fo:
        fo.write(b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 0, 0, 0, 0)) +
                 _chunk(b"IDAT", z) + _chunk(b"IEND", b""))
