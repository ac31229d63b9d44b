"""Build a tiny synthetic DTU tree and run the reference's OWN DTUDataSet on it (build container only; test infrastructure).

    python oracle/make_golden_dtu.py      # writes tests/golden/dtu_tiny/ (a few small files) + tests/golden/g13_dtu_sample.npz

The tree holds what reference src/data/dtu.py touches for ONE sample: 36 camera files, five 640x512 rectified PNGs (target camera 2
+ the source cameras 30, 10, 6, 35 under light 3), four TransMVSNet depth PNGs and their confidence PNGs (uint16, 512x640; the
reader asserts that size), a one-line scan list.  All files are written by this script from seeded arrays (PNG encoding through
diner_amd.imageio / a 16-bit variant below); the reference class then produces the sample dict that the repo's DTUSamples must reproduce.
"""
import importlib
import os
import struct
import sys
import zlib

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden")
TREE = os.path.join(OUT, "dtu_tiny")


def write_png16(path, a):
    """(H,W) uint16 -> 16-bit greyscale PNG."""
    H, W = a.shape
    raw = np.concatenate([np.zeros((H, 1), np.uint8), a.astype(">u2").view(np.uint8).reshape(H, 2 * W)], axis=1).tobytes()

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, 16, 0, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 9)) + chunk(b"IEND", b""))


def build_tree():
    from diner_amd.png import write_png
    from diner_amd.synthetic import look_at_extrinsics
    g = np.random.default_rng(13)
    os.makedirs(os.path.join(TREE, "Cameras", "train"), exist_ok=True)
    for i in range(36):
        th = np.deg2rad(-50 + 100 * i / 35.0)
        E = look_at_extrinsics((600 * np.sin(th), -80 + 4 * i, -600 * np.cos(th))).numpy().astype(np.float64)   # DTU units (mm)
        K = np.array([[361.5 + i * 0.1, 0, 82.9], [0, 360.4, 66.4], [0, 0, 1]])
        with open(os.path.join(TREE, "Cameras", "train", f"{i:08d}_cam.txt"), "w") as f:
            f.write("extrinsic\n" + "\n".join(" ".join(f"{v:.6f}" for v in row) for row in E) + "\n\nintrinsic\n" +
                    "\n".join(" ".join(f"{v:.6f}" for v in row) for row in K) + "\n\n425.0 2.5\n")
    scan = "scan_tiny"
    os.makedirs(os.path.join(TREE, "Rectified", scan + "_train"), exist_ok=True)
    os.makedirs(os.path.join(TREE, "Depths", scan), exist_ok=True)
    yy, xx = np.mgrid[0:512, 0:640]
    for cam in (2, 30, 10, 6, 35):
        # smooth low-entropy image (compresses to a few KB) + a little seeded texture
        img = np.stack([(xx * 255 // 639 + cam * 3) % 256, (yy * 255 // 511 + cam * 5) % 256,
                        ((xx // 16 + yy // 16 + cam) % 2) * 180 + 20], -1).astype(np.uint8)
        img[::37, ::41] = g.integers(0, 256, size=img[::37, ::41].shape, dtype=np.uint8)
        write_png(os.path.join(TREE, "Rectified", scan + "_train", f"rect_{cam + 1:03d}_3_r5000.png"), img, level=9)
        if cam != 2:
            depth_mm = 500 + 150 * np.sin(xx / 90.0 + cam) * np.cos(yy / 70.0) + (cam % 7)
            depth_mm[:40, :60] = 0                                        # background: no depth
            depth_mm[200:230, 300:340] = 0
            d16 = np.clip(np.round(depth_mm * (0.7 / 872.0) / 1e-4), 0, 65535).astype(np.uint16)      # as TransMVSNet writes it
            conf = np.clip(0.5 + 0.45 * np.cos(xx / 50.0) * np.sin(yy / 45.0 + cam), 0, 1)
            c16 = np.round(conf * (0.7 / 872.0) / 1e-4 * 100).astype(np.uint16)                        # goes through read_depth too
            write_png16(os.path.join(TREE, "Depths", scan, f"depth_map_{cam:04d}_TransMVSNet.png"), d16)
            write_png16(os.path.join(TREE, "Depths", scan, f"depth_map_{cam:04d}_TransMVSNet_conf.png"), c16)
    with open(os.path.join(TREE, "scan_list.txt"), "w") as f:
        f.write(scan + "\n" + scan + "\n")          # two entries: the reference's np.loadtxt needs more than one line
    return scan


def main():
    scan = build_tree()
    from oracle.ref_import import import_reference
    ns = import_reference()
    sys.modules.update(ns._modules)
    tv = sys.modules["torchvision.transforms"]
    tvf = sys.modules["torchvision.transforms.functional"]

    class InterpolationMode:
        NEAREST = "nearest"

    def pil_to_tensor(pic):                       # torchvision.transforms.functional.pil_to_tensor: (C,H,W), dtype of the image
        a = np.asarray(pic)
        if a.dtype == np.int32 or str(pic.mode).startswith("I"):
            a = a.astype(np.int32)
        t = torch.from_numpy(np.ascontiguousarray(a))
        return t[None] if t.dim() == 2 else t.permute(2, 0, 1)

    def resize(img, size, interpolation=None):    # tensor path of torchvision's resize with NEAREST = F.interpolate(mode="nearest")
        assert interpolation == InterpolationMode.NEAREST
        return torch.nn.functional.interpolate(img, size, mode="nearest")
    tv.InterpolationMode, tvf.pil_to_tensor, tvf.resize = InterpolationMode, pil_to_tensor, resize
    sys.path.insert(0, "/root/reference")
    dtu = importlib.import_module("src.data.dtu")
    for k in [k for k in sys.modules if k == "src" or k.startswith("src.")]:
        del sys.modules[k]
    sys.path.remove("/root/reference")
    cwd = os.getcwd()
    os.makedirs("/tmp/dtu_ref_cwd/assets/data_splits/dtu", exist_ok=True)
    for n in ("dtu_val_all.txt", "dtu_train_all.txt"):
        with open(f"/tmp/dtu_ref_cwd/assets/data_splits/dtu/{n}", "w") as f:
            f.write(scan + "\n" + scan + "\n")
    os.chdir("/tmp/dtu_ref_cwd")
    try:
        ds = dtu.DTUDataSet(TREE, "val")
        idx = (0 * ds.ncams + 2) * ds.nlights + 3             # scan 0, target camera index 2, light 3
        s = ds[idx]
    finally:
        os.chdir(cwd)
    from diner_amd.datasets import DTUSamples
    mine = DTUSamples(TREE, "val", scan_list=os.path.join(TREE, "scan_list.txt"))
    assert len(mine) == len(ds)
    m = mine[idx]
    out = {}
    for k, v in s.items():
        if torch.is_tensor(v):
            eq = torch.equal(v, m[k]) and v.dtype == m[k].dtype
            print(f"  {k:18s} {tuple(v.shape)} {v.dtype}  identical={eq}")
            assert eq, k
            out[k] = v.numpy()
        else:
            assert v == m[k], k
            out[k] = np.array(v)
    np.savez_compressed(os.path.join(OUT, "g13_dtu_sample.npz"), idx=idx, n=len(ds), znear=ds.znear, zfar=ds.zfar, **out)
    print("reference sample dict reproduced bit for bit; fixture written")


if __name__ == "__main__":
    main()
