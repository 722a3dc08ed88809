"""Tiny PNG writer (zlib only) so test artifacts can be eyeballed, like the reference's dump_png."""
import struct
import zlib

import numpy as np


def write_png(path, rgba8: np.ndarray):
    h, w, c = rgba8.shape
    assert c == 4
    raw = b"".join(b"\x00" + rgba8[y].tobytes() for y in range(h))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0))
                + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))
