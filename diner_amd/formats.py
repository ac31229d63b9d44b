"""On-disk formats feeding the renderer: host-side readers, numpy + zlib only.

* PFM depth maps                      (reference src/util/io.py:4-39: header 'Pf' / 'PF', '<w> <h>', signed scale =
                                       endianness, rows bottom-up)
* DTU `*_cam.txt` camera files        (src/data/dtu.py:143-155: extrinsics lines [1,5), intrinsics lines [7,10),
                                       depth_min and depth_interval on line 11, depth_max = min + 192 * interval)
* TransMVSNet uint16 PNG depth / confidence predictions (value x 1e-4; src/data/dtu.py:104-108: DTU depths are further
                                       divided by 0.7 / 872, the scale used when TransMVSNet was trained)
* confidence -> depth standard deviation: one linear law per dataset (dtu.py:68-70; facescape.py:50-52; multiface.py:305-310
                                       with its clamp at 0 and sigma = 0 where there is no depth)

The reference reads PNGs through PIL and resizes with torchvision; neither exists here, so the PNG reader below
implements the PNG specification itself (8 / 16-bit grey, grey+alpha, RGB, RGBA; all five row filters; no interlace).
"""
import re
import struct
import zlib

import numpy as np


# ---- PFM ------------------------------------------------------------------------------------------------------------
def read_pfm(path):
    """-> (array float32 (H,W) or (H,W,3), rows top-down; scale)."""
    with open(path, "rb") as f:
        header = f.readline().decode("utf-8").rstrip()
        if header not in ("PF", "Pf"):
            raise ValueError("not a PFM file")
        m = re.match(r"^(\d+)\s(\d+)\s$", f.readline().decode("utf-8"))
        if not m:
            raise ValueError("malformed PFM header")
        w, h = int(m.group(1)), int(m.group(2))
        scale = float(f.readline().rstrip())
        endian = "<" if scale < 0 else ">"
        data = np.frombuffer(f.read(), dtype=endian + "f4")
    shape = (h, w, 3) if header == "PF" else (h, w)
    return np.flipud(data.reshape(shape)).astype(np.float32), abs(scale)


def write_pfm(path, image, scale=1.0):
    """float32 (H,W) or (H,W,3) -> little-endian PFM (rows bottom-up, negative scale)."""
    a = np.asarray(image, dtype=np.float32)
    if a.ndim not in (2, 3) or (a.ndim == 3 and a.shape[2] != 3):
        raise ValueError("PFM images are (H,W) or (H,W,3)")
    with open(path, "wb") as f:
        f.write(b"PF\n" if a.ndim == 3 else b"Pf\n")
        f.write(f"{a.shape[1]} {a.shape[0]}\n".encode())
        f.write(f"{-abs(scale)}\n".encode())
        f.write(np.flipud(a).astype("<f4").tobytes())


# ---- DTU camera files ---------------------------------------------------------------------------------------------
def read_dtu_cam(path, n_depth_planes=192):
    """-> intrinsics (3,3) float32, extrinsics (4,4) float32 world->camera, [depth_min, depth_max]."""
    with open(path) as f:
        lines = [line.rstrip() for line in f.readlines()]
    extr = np.array(" ".join(lines[1:5]).split(), dtype=np.float32).reshape(4, 4)
    intr = np.array(" ".join(lines[7:10]).split(), dtype=np.float32).reshape(3, 3)
    tok = lines[11].split()
    dmin = float(tok[0])
    return intr, extr, [dmin, dmin + float(tok[1]) * n_depth_planes]


# ---- PNG ------------------------------------------------------------------------------------------------------------
def read_png(path):
    """8 / 16-bit non-interlaced PNG -> uint8 / uint16 array (H,W) or (H,W,C)."""
    data = open(path, "rb").read()
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError("not a PNG file")
    pos, idat, hdr = 8, [], None
    while pos < len(data):
        n, tag = struct.unpack(">I", data[pos:pos + 4])[0], data[pos + 4:pos + 8]
        body = data[pos + 8:pos + 8 + n]
        if zlib.crc32(tag + body) & 0xffffffff != struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0]:
            raise ValueError(f"PNG chunk {tag!r}: CRC mismatch")
        if tag == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif tag == b"IDAT":
            idat.append(body)
        elif tag == b"IEND":
            break
        pos += 12 + n
    W, H, depth, color, _, _, interlace = hdr
    if depth not in (8, 16) or interlace != 0 or color not in (0, 2, 4, 6):
        raise ValueError(f"unsupported PNG (bit depth {depth}, colour type {color}, interlace {interlace})")
    ch = {0: 1, 2: 3, 4: 2, 6: 4}[color]
    bpp = ch * depth // 8                                   # bytes per pixel = the filters' "left" distance
    stride = W * bpp
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), np.uint8).reshape(H, 1 + stride)
    out = np.zeros((H, stride), np.uint8)
    prev = np.zeros(stride, np.int32)
    for y in range(H):
        ft, line = int(raw[y, 0]), raw[y, 1:].astype(np.int32)
        if ft == 0:
            cur = line
        elif ft == 2:                                        # Up
            cur = (line + prev) & 255
        elif ft in (1, 3, 4):                                # Sub / Average / Paeth: sequential in x
            cur = np.zeros(stride, np.int32)
            for x in range(stride):
                a = cur[x - bpp] if x >= bpp else 0
                b = prev[x]
                if ft == 1:
                    pred = a
                elif ft == 3:
                    pred = (a + b) >> 1
                else:
                    c = prev[x - bpp] if x >= bpp else 0
                    p = a + b - c
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cur[x] = (line[x] + pred) & 255
        else:
            raise ValueError(f"PNG filter type {ft}")
        out[y] = cur
        prev = cur
    if depth == 16:
        img = out.reshape(H, W, ch, 2).astype(np.uint16)
        img = (img[..., 0] << 8) | img[..., 1]               # big-endian samples
    else:
        img = out.reshape(H, W, ch)
    return img[..., 0] if ch == 1 else img


TRANSMVSNET_SCALE = 1e-4          # uint16 PNG value -> metres-like units (train.py:152-191, utils.py:21-23)
DTU_DEPTH_RESCALE = 0.7 / 872.0   # dtu.py:106: undo the scale TransMVSNet was trained with


def read_transmvsnet_png(path, dtu_rescale=False):
    """uint16 PNG written by the TransMVSNet stage -> float32 (H,W): value * 1e-4 (depth: [/ (0.7/872) on DTU]; confidence
    maps use the same container without the DTU factor)."""
    img = read_png(path)
    if img.ndim == 3:
        img = img[..., 0]
    out = img.astype(np.float32) * np.float32(TRANSMVSNET_SCALE)
    if dtu_rescale:
        out = out / np.float32(DTU_DEPTH_RESCALE)
    return out


CONF_TO_STD = {"dtu": (-2.5679e-2, 3.2818e-2),          # dtu.py:68-70
               "facescape": (-1.582e-2, 1.649e-2),      # facescape.py:50-52
               "multiface": (-1.582e-2, 1.649e-2)}      # multiface.py:309, then clipped at 0 and zeroed where depth == 0


def conf_to_std(conf, law="dtu", depth=None):
    """TransMVSNet confidence in [0,1] -> depth standard deviation a * conf + b with the data set's coefficients.  The Multiface law
    also clamps at 0 and returns 0 where `depth` (same shape, optional) is 0 (multiface.py:309-310)."""
    a, b = CONF_TO_STD[law]
    std = a * conf + b
    if law == "multiface":
        std = std.clip(min=0) if hasattr(std, "clip") else max(std, 0.0)
        if depth is not None:
            std = std.clone() if hasattr(std, "clone") else np.array(std, copy=True)
            std[depth == 0] = 0
    return std
