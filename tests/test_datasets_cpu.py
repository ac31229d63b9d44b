"""CPU: batch-dict assembly from the on-disk DTU layout (row f4) against the sample dict the reference's OWN DTUDataSet produced for
the tiny tree under tests/golden/dtu_tiny (oracle/make_golden_dtu.py ran it in the build container; it asserts bit-identity there)."""
import os

import numpy as np
import pytest
import torch

from tests.helpers import GOLD, load

TREE = os.path.join(GOLD, "dtu_tiny")


def _dataset():
    from diner_amd.datasets import DTUSamples
    return DTUSamples(TREE, "val", scan_list=os.path.join(TREE, "scan_list.txt"))


def test_dtu_sample_matches_reference_dataset():
    g = load("g13_dtu_sample.npz")
    ds = _dataset()
    assert len(ds) == int(g["n"]) == 2 * 36 * 7
    assert abs(ds.znear - float(g["znear"])) < 1e-12 and abs(ds.zfar - float(g["zfar"])) < 1e-12
    s = ds[int(g["idx"])]
    try:
        import PIL  # noqa: F401
        have_pil = True
    except ImportError:
        have_pil = False
    for k in g.files:
        if k in ("idx", "n", "znear", "zfar"):
            continue
        want = g[k]
        if torch.is_tensor(s[k]):
            got = s[k].numpy()
            assert got.shape == want.shape and got.dtype == want.dtype, k
            if k in ("target_rgb", "src_rgbs") and not have_pil:
                assert np.abs(got - want).max() < 0.1          # 2x2 average instead of PIL's bicubic resampler
            else:
                assert np.array_equal(got, want), k
        else:
            assert str(s[k]) == str(want), k
    # depth holes carry through: mask 0 and depth 0 in the same places, std map from the confidence law
    assert torch.equal(s["src_alphas"] == 0, s["src_depths"] == 0) and (s["src_alphas"] == 0).any()
    assert float(s["src_depth_stds"].max()) <= 3.2818e-2 + 1e-6


def test_collate_and_encode_args():
    from diner_amd.datasets import collate, encode_args
    ds = _dataset()
    b = collate([ds[17], ds[(1 * 36 + 2) * 7 + 3]])          # the tree holds the files of one (camera, light) only: both list entries
    assert b["src_rgbs"].shape == (2, 4, 3, 256, 320) and b["target_extrinsics"].shape == (2, 4, 4)
    assert b["sample_name"] == ["scan_tiny-2", "scan_tiny-2"]
    a = encode_args(b)
    assert set(a) == {"images", "depths", "depths_std", "extrinsics", "intrinsics"}
    assert a["depths"].shape == (2, 4, 1, 256, 320) and a["intrinsics"].shape == (2, 4, 3, 3)
    sweep = ds.get_cam_sweep_extrinsics(5)
    assert sweep.shape == (5, 4, 4) and torch.isfinite(sweep).all()


def test_missing_tree_is_an_error(tmp_path):
    from diner_amd.datasets import DTUSamples
    with pytest.raises(FileNotFoundError):
        DTUSamples(str(tmp_path / "nope"), "val", scan_list=["x"])


def test_facescape_samples_match_reference_dataset(tmp_path):
    """FacescapeSamples on the tiny tree under tests/golden/facescape_tiny against what the reference's OWN FacescapeDataSet produced
    there (oracle/make_golden_facescape.py): the view-selection list entry for entry, three sample dicts bit for bit, the sweep."""
    import json
    import shutil
    from diner_amd.datasets import FacescapeSamples, collate, encode_args
    g = load("g14_facescape.npz")
    tree = os.path.join(GOLD, "facescape_tiny")
    shutil.copy(os.path.join(tree, "splits", "publishable_list_v1.txt"), tmp_path)      # the list cache is written next to it
    ds = FacescapeSamples(tree, "val", split_dir=str(tmp_path))
    want_metas = json.loads(str(g["metas_json"]))
    assert len(ds) == int(g["n"]) == len(want_metas) > 0
    assert json.loads(json.dumps(ds.metas)) == want_metas
    # the cache file has the reference's name and loads back to the same list
    assert (tmp_path / "val_45_30_40.txt").exists()
    assert FacescapeSamples(tree, "val", split_dir=str(tmp_path)).metas == want_metas
    for j, i in enumerate(g["picks"].tolist()):
        s = ds[i]
        keys = [k[len(f"s{j}_"):] for k in g.files if k.startswith(f"s{j}_")]
        assert set(keys) == set(s.keys())
        for k in keys:
            want = g[f"s{j}_{k}"]
            if torch.is_tensor(s[k]):
                got = s[k].numpy()
                assert got.shape == want.shape and got.dtype == want.dtype and np.array_equal(got, want), k
            else:
                assert s[k] == want.item(), k
    sw = ds.get_cam_sweep_extrinsics(7, int(g["picks"][1]))
    assert sw.shape == (7, 4, 4) and np.abs(sw.numpy() - g["sweep"]).max() < 1e-6
    # the dict feeds PixelNeRF.encode like the DTU one
    b = collate([ds[0], ds[1]])
    a = encode_args(b)
    assert a["images"].shape == (2, 4, 3, 24, 32) and a["depths_std"].shape == (2, 4, 1, 24, 32) and a["extrinsics"].shape == (2, 4, 4, 4)
    assert ds.znear == 1.0 and ds.zfar == 2.5


def test_multiface_samples_match_reference_dataset(tmp_path):
    """MultifaceSamples on tests/golden/multiface_tiny against the reference's OWN MultiFaceDataset (oracle/make_golden_multiface.py):
    sample list, one dict with the constant sigma and one with the confidence law (clamped at 0, 0 where there is no depth), the sweep."""
    import json
    import shutil
    from diner_amd.datasets import MultifaceSamples, collate, encode_args
    g = load("g15_multiface.npz")
    tree = os.path.join(GOLD, "multiface_tiny")
    shutil.copy(os.path.join(tree, "splits", "tiny_subset.json"), tmp_path)
    kw = dict(downsample=1, split_config=str(tmp_path / "tiny_subset.json"), split_dir=str(tmp_path))
    ds = MultifaceSamples(tree, "val", **kw)
    ds_c = MultifaceSamples(tree, "val", depth_std_suffix="_conf.png", **kw)          # loads the list the first one cached
    want_metas = json.loads(str(g["metas_json"]))
    assert len(ds) == int(g["n"]) == len(want_metas) > 0 and (tmp_path / "val_tiny_subset.txt").exists()
    assert json.loads(json.dumps(ds.metas)) == want_metas and ds_c.metas == want_metas
    picks = g["picks"].tolist()
    for j, s in enumerate((ds[picks[0]], ds_c[picks[1]])):
        keys = [k[len(f"s{j}_"):] for k in g.files if k.startswith(f"s{j}_")]
        assert set(keys) == set(s.keys())
        for k in keys:
            want = g[f"s{j}_{k}"]
            if torch.is_tensor(s[k]):
                got = s[k].numpy()
                assert got.shape == want.shape and got.dtype == want.dtype and np.array_equal(got, want), k
            else:
                assert s[k] == want.item(), k
    std = ds_c[picks[1]]["src_depth_stds"]
    assert float(std.min()) == 0.0 and bool((std[ds_c[picks[1]]["src_depths"] == 0] == 0).all())
    sw = ds.get_cam_sweep_extrinsics(6, picks[0])
    assert sw.shape == (6, 4, 4) and np.abs(sw.numpy() - g["sweep"]).max() < 1e-6
    # filters act on the loaded list  (the down-sampling branch -- torchvision's resize, restated with F.interpolate -- is not
    # pinned: torchvision is absent from the build container, and the tiny images are already the multiples of 32 it rounds to)
    assert len(MultifaceSamples(tree, "val", target_filter=[want_metas[0]["target_id"]], **kw)) < len(ds)
    a = encode_args(collate([ds[0], ds[1]]))
    assert a["images"].shape == (2, 4, 3, 32, 64) and a["depths"].shape == (2, 4, 1, 32, 64)
