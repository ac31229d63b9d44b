"""Row f4: host-side readers of the on-disk formats that feed the renderer (diner_amd/formats.py)."""
import os
import struct
import zlib

import numpy as np
import pytest

from diner_amd import formats as F

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "formats")


def test_pfm_against_reference_reader(tmp_path):
    """Files in the three header / endianness variants: same arrays and scales as the reference's read_pfm returned when
    the fixtures were made (oracle/make_golden_formats.py); write_pfm round trip."""
    exp = np.load(os.path.join(GOLD, "pfm_expected.npz"))
    for name in ("grey_le", "grey_be", "color_le"):
        data, scale = F.read_pfm(os.path.join(GOLD, name + ".pfm"))
        assert data.dtype == np.float32 and np.array_equal(data, exp[name])
        assert scale == float(exp[name + "_scale"])
    a = np.random.default_rng(1).normal(size=(6, 4)).astype(np.float32)
    F.write_pfm(str(tmp_path / "a.pfm"), a, scale=3.0)
    b, s = F.read_pfm(str(tmp_path / "a.pfm"))
    assert np.array_equal(a, b) and s == 3.0
    with pytest.raises(ValueError):
        (tmp_path / "bad.pfm").write_bytes(b"P5\n1 1\n-1\n\x00\x00\x00\x00")
        F.read_pfm(str(tmp_path / "bad.pfm"))


def test_dtu_cam_file(tmp_path):
    """The MVSNet camera text format of DTU (dtu.py:143-155)."""
    txt = """extrinsic
0.970263 0.00747983 0.241939 -191.02
-0.0147429 0.999493 0.0282234 3.28832
-0.241605 -0.030951 0.969881 22.5401
0.0 0.0 0.0 1.0

intrinsic
2892.33 0 823.205
0 2883.18 619.071
0 0 1

425 2.5
"""
    p = tmp_path / "00000000_cam.txt"
    p.write_text(txt)
    K, E, (dmin, dmax) = F.read_dtu_cam(str(p))
    assert K.dtype == np.float32 and E.dtype == np.float32 and K.shape == (3, 3) and E.shape == (4, 4)
    assert K[0, 0] == np.float32(2892.33) and K[1, 2] == np.float32(619.071) and E[0, 3] == np.float32(-191.02)
    assert np.array_equal(E[3], np.array([0, 0, 0, 1], np.float32))
    assert dmin == 425.0 and dmax == 425.0 + 2.5 * 192


def _png(path, img, filters):
    """Test-side PNG encoder that applies a chosen filter type per row (to exercise every reconstruction branch)."""
    a = np.asarray(img)
    H, W = a.shape[:2]
    ch = 1 if a.ndim == 2 else a.shape[2]
    depth = 16 if a.dtype == np.uint16 else 8
    if depth == 16:
        by = np.stack([(a >> 8).astype(np.uint8), (a & 255).astype(np.uint8)], axis=-1).reshape(H, -1)
    else:
        by = a.reshape(H, -1)
    bpp = ch * depth // 8
    rows, prev = [], np.zeros(by.shape[1], np.int32)
    for y in range(H):
        cur, ft = by[y].astype(np.int32), filters[y % len(filters)]
        left = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]])
        ul = np.concatenate([np.zeros(bpp, np.int32), prev[:-bpp]])
        if ft == 0:
            pred = 0
        elif ft == 1:
            pred = left
        elif ft == 2:
            pred = prev
        elif ft == 3:
            pred = (left + prev) >> 1
        else:
            p = left + prev - ul
            pa, pb, pc = abs(p - left), abs(p - prev), abs(p - ul)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, ul))
        rows.append(bytes([ft]) + ((cur - pred) & 255).astype(np.uint8).tobytes())
        prev = cur

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)
    color = {1: 0, 2: 4, 3: 2, 4: 6}[ch]
    raw = zlib.compress(b"".join(rows))
    with open(path, "wb") as f:      # two IDAT chunks on purpose
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, depth, color, 0, 0, 0)) +
                chunk(b"IDAT", raw[:len(raw) // 2]) + chunk(b"IDAT", raw[len(raw) // 2:]) + chunk(b"IEND", b""))


def test_png_reader_all_filters_and_depths(tmp_path):
    g = np.random.default_rng(3)
    for dtype, shape in ((np.uint16, (9, 11)), (np.uint8, (8, 5, 3)), (np.uint16, (6, 7, 4)), (np.uint8, (5, 5))):
        hi = 65536 if dtype == np.uint16 else 256
        img = g.integers(0, hi, size=shape).astype(dtype)
        p = str(tmp_path / "t.png")
        _png(p, img, filters=[0, 1, 2, 3, 4])
        got = F.read_png(p)
        assert got.dtype == dtype and np.array_equal(got, img)
    from diner_amd import imageio
    a = g.integers(0, 256, size=(4, 6, 3), dtype=np.uint8)
    imageio.write_png(str(tmp_path / "w.png"), a)
    assert np.array_equal(F.read_png(str(tmp_path / "w.png")), a)       # the package's own writer
    with pytest.raises(ValueError):
        (tmp_path / "x.png").write_bytes(b"notapng")
        F.read_png(str(tmp_path / "x.png"))


def test_transmvsnet_depth_and_confidence(tmp_path):
    """uint16 PNG x 1e-4 (/ (0.7/872) for DTU depths, dtu.py:104-108) and the confidence -> std law (dtu.py:68-70)."""
    depth_u16 = np.array([[0, 1, 7000], [65535, 12345, 8720]], np.uint16)
    _png(str(tmp_path / "d.png"), depth_u16, filters=[4])
    d = F.read_transmvsnet_png(str(tmp_path / "d.png"), dtu_rescale=True)
    want = depth_u16.astype(np.float32) * np.float32(1e-4) / np.float32(0.7 / 872.0)
    assert d.dtype == np.float32 and np.array_equal(d, want)
    c = F.read_transmvsnet_png(str(tmp_path / "d.png"))
    assert np.array_equal(c, depth_u16.astype(np.float32) * np.float32(1e-4))
    conf = np.array([0.0, 0.3, 1.0])
    assert np.allclose(F.conf_to_std(conf), [3.2818e-2, 3.2818e-2 - 0.3 * 2.5679e-2, 3.2818e-2 - 2.5679e-2], atol=0, rtol=1e-15)
    # the Facescape / Multiface coefficients (facescape.py:50-52, multiface.py:309-310: clamp at 0, sigma 0 where there is no depth)
    assert np.allclose(F.conf_to_std(conf, "facescape"), [1.649e-2, 1.649e-2 - 0.3 * 1.582e-2, 1.649e-2 - 1.582e-2], atol=0, rtol=1e-15)
    big = np.array([0.0, 1.0, 1.2], dtype=np.float64)
    assert np.array_equal(F.conf_to_std(big, "multiface", depth=np.array([0.0, 1.0, 1.0])),
                          np.array([0.0, 1.649e-2 - 1.582e-2, 0.0]))
