"""Build a tiny synthetic Multiface tree and run the reference's OWN MultiFaceDataset on it (build container only; test infrastructure).

    python oracle/make_golden_multiface.py   # writes tests/golden/multiface_tiny/ (small files) + tests/golden/g15_multiface.npz

One subject, one sequence, two frames, 12 cameras (KRT file, mm), 32x64 RGB images, 8-bit masks (with partly covered pixels), 16-bit
depth and confidence PNGs, a split file with four ideal reference centres.  The reference class produces its cached sample list, sample
dicts (with and without a confidence suffix) and a sweep; the repo's MultifaceSamples must reproduce the list exactly and the tensors
bit for bit.  Image sizes are multiples of 32 and downsample = 1, so the reference's torchvision `resize` branch (absent here) is not
taken; MultifaceSamples implements it with F.interpolate and says so.
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
TREE = os.path.join(OUT, "multiface_tiny")
SPLITS = os.path.join(TREE, "splits")
SUBJ, SEQ, FRAMES = "m--tiny--0001", "E001_Neutral", ("000102", "000105")


def cam_centre(az, el, r=1000.0):
    a, e = np.deg2rad(az), np.deg2rad(el)
    return r * np.array([np.sin(a) * np.cos(e), np.sin(e), -np.cos(a) * np.cos(e)])


def build_tree():
    from diner_amd.png import write_png
    from diner_amd.synthetic import look_at_extrinsics
    from oracle.make_golden_dtu import write_png16
    g = np.random.default_rng(15)
    os.makedirs(SPLITS, exist_ok=True)
    os.makedirs(os.path.join(TREE, SUBJ), exist_ok=True)
    H, W = 32, 64
    yy, xx = np.mgrid[0:H, 0:W]
    names, krt = [], []
    k = 0
    for el in (-25.0, 0.0, 25.0):
        for az in (-40.0, -13.0, 13.0, 40.0):
            c = cam_centre(az + 0.8 * k, el + 0.5 * k)
            E = look_at_extrinsics(tuple(c)).numpy().astype(np.float64)[:3]
            name = str(400002 + 7 * k)
            names.append(name)
            K = [[900.0 + k, 0.0, 32.0], [0.0, 905.0, 16.0], [0.0, 0.0, 1.0]]
            krt.append(name + "\n" + "\n".join(" ".join(f"{v:.6f}" for v in row) for row in K) + "\n" +
                       " ".join(f"{v:.6f}" for v in (0.01 * k, -0.02, 0.0, 0.0, 0.0)) + "\n" +
                       "\n".join(" ".join(f"{v:.6f}" for v in row) for row in E) + "\n\n")
            for fr in FRAMES:
                f = int(fr)
                for sub in ("images", "masks", "depths"):
                    os.makedirs(os.path.join(TREE, SUBJ, sub, SEQ, name), exist_ok=True)
                img = np.stack([(xx * 3 + k * 13 + f) % 256, (yy * 7 + k * 3) % 256, ((xx // 4 + yy // 4 + k) % 2) * 150 + 40], -1).astype(np.uint8)
                img[::6, ::9] = g.integers(0, 256, size=img[::6, ::9].shape, dtype=np.uint8)
                write_png(os.path.join(TREE, SUBJ, "images", SEQ, name, fr + ".png"), img, level=9)
                r2 = (xx - 32) ** 2 + 3 * (yy - 16) ** 2
                mask = np.where(r2 < 500 + 10 * k, 255, np.where(r2 < 700 + 10 * k, 128, 0)).astype(np.uint8)
                write_png(os.path.join(TREE, SUBJ, "masks", SEQ, name, fr + ".png"), mask, level=9)
                depth = 0.95 + 0.12 * np.sin(xx / 9.0 + k) * np.cos(yy / 6.0 + f)
                depth[r2 >= 700 + 10 * k] = 0
                conf = np.clip(0.55 + 0.6 * np.cos(xx / 5.0) * np.sin(yy / 4.0 + k), 0, 1.2)     # > 1.04 exercises the clamp at 0
                write_png16(os.path.join(TREE, SUBJ, "depths", SEQ, name, fr + ".png"), np.round(depth / 1e-4).astype(np.uint16))
                write_png16(os.path.join(TREE, SUBJ, "depths", SEQ, name, fr + "_conf.png"), np.round(conf / 1e-4).astype(np.uint16))
            k += 1
    with open(os.path.join(TREE, SUBJ, "KRT"), "w") as f:
        f.write("".join(krt))
    corners = [cam_centre(-38, -24), cam_centre(38, -24), cam_centre(38, 27), cam_centre(-38, 27)]
    part = dict(subjects=[SUBJ], sequences=[SEQ], ref_centers=[[round(float(v), 3) for v in c] for c in corners])
    with open(os.path.join(SPLITS, "tiny_subset.json"), "w") as f:
        json.dump(dict(train=part, val=part), f)


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
    sys.path.insert(0, "/root/reference")
    mf = importlib.import_module("src.data.multiface")
    for k in [k for k in sys.modules if k == "src" or k.startswith("src.")]:
        del sys.modules[k]
    sys.path.remove("/root/reference")
    cwd = os.getcwd()
    work = "/tmp/multiface_ref_cwd"
    shutil.rmtree(work, ignore_errors=True)
    os.makedirs(os.path.join(work, "assets/data_splits/multiface"))
    cfg = os.path.join(work, "assets/data_splits/multiface/tiny_subset.json")
    shutil.copy(os.path.join(SPLITS, "tiny_subset.json"), cfg)
    os.chdir(work)
    try:
        from pathlib import Path
        ds = mf.MultiFaceDataset(TREE, "val", downsample=1, split_config=Path(cfg))
        ds_c = mf.MultiFaceDataset(TREE, "val", downsample=1, split_config=Path(cfg), depth_std_suffix="_conf.png")
        ref_metas = ds.metas
        picks = [0, len(ds) - 1]
        ref_samples = [ds[picks[0]], ds_c[picks[1]]]
        ref_sweep = ds.get_cam_sweep_extrinsics(6, picks[0])
    finally:
        os.chdir(cwd)
    from diner_amd.datasets import MultifaceSamples
    cache = os.path.join(SPLITS, "val_tiny_subset.txt")
    if os.path.exists(cache):
        os.remove(cache)
    kw = dict(downsample=1, split_config=os.path.join(SPLITS, "tiny_subset.json"), split_dir=SPLITS)
    mine, mine_c = MultifaceSamples(TREE, "val", **kw), MultifaceSamples(TREE, "val", depth_std_suffix="_conf.png", **kw)
    os.remove(cache)                                           # the cache file is not part of the fixture (tests rebuild it)
    assert len(mine) == len(ds) > 0, (len(mine), len(ds))
    assert json.loads(json.dumps(mine.metas)) == json.loads(json.dumps(ref_metas)), "sample list differs"
    out = {"n": len(ds), "picks": np.array(picks), "metas_json": np.array(json.dumps(ref_metas))}
    for j, (s, m) in enumerate(zip(ref_samples, [mine[picks[0]], mine_c[picks[1]]])):
        assert set(m.keys()) == set(s.keys()), (set(m.keys()) ^ set(s.keys()))
        for k, v in s.items():
            if torch.is_tensor(v):
                eq = torch.equal(v, m[k]) and v.dtype == m[k].dtype
                print(f"  sample {picks[j]:3d} {k:18s} {tuple(v.shape)} {v.dtype}  identical={eq}")
                assert eq, k
                out[f"s{j}_{k}"] = v.numpy()
            else:
                assert v == m[k] and type(v) is type(m[k]), (k, v, m[k])
                out[f"s{j}_{k}"] = np.array(v)
    sw = mine.get_cam_sweep_extrinsics(6, picks[0])
    err = (sw - ref_sweep).abs().max().item()
    print(f"  sweep (6,4,4): max |diff| {err:.2e}")
    assert err < 1e-6
    out["sweep"] = ref_sweep.numpy()
    np.savez_compressed(os.path.join(OUT, "g15_multiface.npz"), **out)
    print(f"reference sample list ({len(ds)} entries), 2 sample dicts (constant sigma / confidence law) and the sweep reproduced; fixture written")


if __name__ == "__main__":
    main()
