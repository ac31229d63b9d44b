"""Build a tiny synthetic Facescape tree and run the reference's OWN FacescapeDataSet on it (build container only; test infrastructure).

    python oracle/make_golden_facescape.py   # writes tests/golden/facescape_tiny/ (a few small files) + tests/golden/g14_facescape.npz

The tree holds one scan of one validation subject (122/01): cameras.json with 15 cameras on a cap around the frontal axis (ids are
strings and sort as strings, as in the real captures), a landmark file, and per view an RGBA PNG (24x32), a 16-bit depth PNG and its
16-bit confidence PNG.  All files are written by this script from seeded arrays.  The reference class then produces (a) its cached
sample list ("metas": four source-candidate lists + a target per entry), (b) sample dicts, (c) a sweep path; the repo's
FacescapeSamples must reproduce the list exactly and the tensors bit for bit.
"""
import importlib
import json
import os
import shutil
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden")
TREE = os.path.join(OUT, "facescape_tiny")
SPLITS = os.path.join(TREE, "splits")          # publishable_list_v1.txt lives here (the reference reads it relative to its cwd)


def build_tree():
    from diner_amd.png import write_png
    from diner_amd.synthetic import look_at_extrinsics
    from oracle.make_golden_dtu import write_png16
    g = np.random.default_rng(14)
    scan = os.path.join(TREE, "122", "01")
    os.makedirs(scan, exist_ok=True)
    os.makedirs(SPLITS, exist_ok=True)
    with open(os.path.join(SPLITS, "publishable_list_v1.txt"), "w") as f:
        f.write("122,212\n")
    np.save(os.path.join(scan, "3dlmks.npy"), g.normal(size=(68, 3)).astype(np.float32))
    cams = {}
    H, W = 24, 32
    yy, xx = np.mgrid[0:H, 0:W]
    k = 0
    for el in (-38.0, 0.0, 38.0):
        for az in (-62.0, -31.0, 0.0, 31.0, 62.0):
            a, e = np.deg2rad(az + 1.5 * k), np.deg2rad(el + 0.7 * k)
            c = 1.7 * np.array([np.sin(a) * np.cos(e), -np.cos(a) * np.cos(e), np.sin(e)])
            E = look_at_extrinsics(tuple(c)).numpy().astype(np.float64)
            cams[str(k)] = dict(extrinsics=[[round(float(v), 6) for v in row] for row in E[:3]],
                                intrinsics=[[40.0 + k, 0.0, 16.0], [0.0, 41.0, 12.0], [0.0, 0.0, 1.0]])
            v = os.path.join(scan, f"view_{k:05d}")
            os.makedirs(v, exist_ok=True)
            rgba = np.stack([(xx * 7 + k * 11) % 256, (yy * 9 + k * 5) % 256, ((xx + yy + k) % 2) * 200 + 30,
                             np.where((xx - 16) ** 2 + (yy - 12) ** 2 < 90 + k, 255, (k * 17) % 120)], -1).astype(np.uint8)
            rgba[::5, ::7, :3] = g.integers(0, 256, size=rgba[::5, ::7, :3].shape, dtype=np.uint8)
            write_png(os.path.join(v, "rgba_colorcalib.png"), rgba, level=9)
            depth = 1.45 + 0.3 * np.sin(xx / 6.0 + k) * np.cos(yy / 5.0) + 0.01 * k
            depth[(xx - 16) ** 2 + (yy - 12) ** 2 >= 90 + k] = 0
            conf = np.clip(0.5 + 0.45 * np.cos(xx / 4.0) * np.sin(yy / 3.0 + k), 0, 1)
            write_png16(os.path.join(v, "depth_TransMVSNet.png"), np.round(depth / 1e-4).astype(np.uint16))
            write_png16(os.path.join(v, "depth_TransMVSNet_conf.png"), np.round(conf / 1e-4).astype(np.uint16))
            k += 1
    with open(os.path.join(scan, "cameras.json"), "w") as f:
        json.dump(cams, f)
    return scan


def main():
    build_tree()
    from oracle.ref_import import import_reference
    ns = import_reference()
    sys.modules.update(ns._modules)
    tvf = sys.modules["torchvision.transforms.functional"]

    def pil_to_tensor(pic):                       # torchvision.transforms.functional.pil_to_tensor: (C,H,W), dtype of the image
        a = np.asarray(pic)
        if a.dtype == np.int32 or str(pic.mode).startswith("I"):
            a = a.astype(np.int32)
        t = torch.from_numpy(np.ascontiguousarray(a))
        return t[None] if t.dim() == 2 else t.permute(2, 0, 1)
    tvf.pil_to_tensor = pil_to_tensor
    if "tqdm" not in sys.modules:
        import types
        tq = types.ModuleType("tqdm")
        tq.tqdm = lambda it, *a, **k: it
        try:
            import tqdm  # noqa: F401
        except ImportError:
            sys.modules["tqdm"] = tq
    sys.path.insert(0, "/root/reference")
    fs = importlib.import_module("src.data.facescape")
    for k in [k for k in sys.modules if k == "src" or k.startswith("src.")]:
        del sys.modules[k]
    sys.path.remove("/root/reference")
    cwd = os.getcwd()
    work = "/tmp/facescape_ref_cwd"
    shutil.rmtree(work, ignore_errors=True)
    os.makedirs(os.path.join(work, "assets/data_splits/facescape"))
    shutil.copy(os.path.join(SPLITS, "publishable_list_v1.txt"), os.path.join(work, "assets/data_splits/facescape"))
    os.chdir(work)
    try:
        ds = fs.FacescapeDataSet(TREE, "val")
        ref_metas = ds.metas
        picks = [0, len(ds) // 2, len(ds) - 1]
        ref_samples = [ds[i] for i in picks]
        ref_sweep = ds.get_cam_sweep_extrinsics(7, picks[1])
    finally:
        os.chdir(cwd)
    for stale in (os.path.join(SPLITS, "val_45_30_40.txt"),):
        if os.path.exists(stale):
            os.remove(stale)
    from diner_amd.datasets import FacescapeSamples
    mine = FacescapeSamples(TREE, "val", split_dir=SPLITS)
    assert len(mine) == len(ds) and len(ds) > 0, (len(mine), len(ds))
    assert json.loads(json.dumps(mine.metas)) == json.loads(json.dumps(ref_metas)), "sample list differs"
    os.remove(os.path.join(SPLITS, "val_45_30_40.txt"))        # the cache file is not part of the fixture (tests rebuild it)
    out = {"n": len(ds), "picks": np.array(picks), "metas_json": np.array(json.dumps(ref_metas))}
    for j, (i, s) in enumerate(zip(picks, ref_samples)):
        m = mine[i]
        assert set(m.keys()) == set(s.keys()), (set(m.keys()) ^ set(s.keys()))
        for k, v in s.items():
            if torch.is_tensor(v):
                eq = torch.equal(v, m[k]) and v.dtype == m[k].dtype
                print(f"  sample {i:3d} {k:18s} {tuple(v.shape)} {v.dtype}  identical={eq}")
                assert eq, k
                out[f"s{j}_{k}"] = v.numpy()
            else:
                assert v == m[k] and type(v) is type(m[k]), (k, v, m[k])
                out[f"s{j}_{k}"] = np.array(v)
    sw = mine.get_cam_sweep_extrinsics(7, picks[1])
    err = (sw - ref_sweep).abs().max().item()
    print(f"  sweep (7,4,4): max |diff| {err:.2e}")
    assert err < 1e-6
    out["sweep"] = ref_sweep.numpy()
    np.savez_compressed(os.path.join(OUT, "g14_facescape.npz"), **out)
    print(f"reference sample list ({len(ds)} entries), 3 sample dicts and the sweep reproduced; fixture written")


if __name__ == "__main__":
    main()
