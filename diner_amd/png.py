"""Dependency-free PNG container (zlib only): the writer of the image output (diner_amd.imageio) and the
reader the tests use.  Kept apart from imageio.py so that fixture generators and data tools can write PNGs WITHOUT the HIP extension
(imageio imports diner_amd.ops, which loads libdiner_hip.so)."""
import struct
import zlib

import numpy as np


def write_png(path, arr, level=1):
    """(H,W,3) or (H,W) uint8 array / tensor -> 8-bit PNG (filter type 0 rows, one IDAT)."""
    a = arr.detach().cpu().numpy() if hasattr(arr, "detach") else np.asarray(arr)
    assert a.dtype == np.uint8 and a.ndim in (2, 3)
    H, W = a.shape[:2]
    ch = 1 if a.ndim == 2 else a.shape[2]
    color = {1: 0, 3: 2, 4: 6}[ch]
    raw = np.concatenate([np.zeros((H, 1), np.uint8), a.reshape(H, W * ch)], axis=1).tobytes()

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, 8, color, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, level)) + chunk(b"IEND", b""))


def read_png(path):
    """Reader for what write_png writes (8-bit, non-interlaced, filter types 0-4) -> uint8 array; used by the tests."""
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, hdr = 8, b"", None
    while pos < len(data):
        n, tag = struct.unpack(">I", data[pos:pos + 4])[0], data[pos + 4:pos + 8]
        body = data[pos + 8:pos + 8 + n]
        assert zlib.crc32(tag + body) & 0xffffffff == struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0]
        if tag == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif tag == b"IDAT":
            idat += body
        pos += 12 + n
    W, H, depth, color, _, _, interlace = hdr
    assert depth == 8 and interlace == 0
    ch = {0: 1, 2: 3, 6: 4}[color]
    rows = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(H, 1 + W * ch)
    assert (rows[:, 0] == 0).all(), "only filter type 0 rows are supported"
    out = rows[:, 1:].reshape(H, W, ch)
    return out[..., 0] if ch == 1 else out
