"""Batch-dict assembly from the on-disk DTU layout: the consumer of diner_amd.formats.

Mirrors what reference src/data/dtu.py::DTUDataSet hands to DINER (`__getitem__`, :183-239): one target view + the four
source views [30, 10, 6, 35] of a scan under one lighting, images downsampled by 0.5, TransMVSNet depth / confidence PNGs
turned into depth + standard-deviation maps, cameras rescaled to the 0.7/872 world scale.  Keys, shapes and dtypes of the
returned dict are the reference's; `collate` adds the batch dimension the way the default DataLoader collation does, so
`nerf.encode(**encode_args(batch))` and `predict_image(...)` run on it directly.

File decoding: depth / confidence maps and camera files go through this package's own readers (numpy + zlib).  The RGB
images are decoded AND downsampled with PIL -- exactly the reference's two calls (`Image.open`, `Image.resize`, dtu.py:79-83);
re-implementing PIL's fixed-point bicubic resampler would be the only way to stay bit-identical without it, and it is host-side
IO outside the hot path.  Without PIL the images are decoded by the package's PNG reader and averaged 2x2 (stated in the
returned dict as `rgb_resample`).

Not built: the Facescape / Multiface classes (same schema, different file trees) and the Lightning DataModule.
"""
import os
from itertools import product
from pathlib import Path

import numpy as np
import torch

from . import formats

SRC_CAM_IDCS = [30, 10, 6, 35]        # dtu.py:48
SCALE_FACTOR = 0.7 / 872.0            # dtu.py:21


class DTUSamples:
    def __init__(self, root, stage="val", scan_list=None, scale_factor=SCALE_FACTOR, downsample=0.5,
                 depth_fname="TransMVSNet"):
        self.data_dir = Path(root)
        if not self.data_dir.exists():
            raise FileNotFoundError(root)
        self.stage, self.scale_factor, self.downsample, self.depth_fname = stage, scale_factor, downsample, depth_fname
        if scan_list is None:          # the reference reads these lists relative to the working directory (dtu.py:131-139)
            scan_list = {"train": "assets/data_splits/dtu/dtu_train_all.txt", "val": "assets/data_splits/dtu/dtu_val_all.txt"}[stage]
        self.scan_list = (np.atleast_1d(np.loadtxt(scan_list, str)) if isinstance(scan_list, (str, os.PathLike))
                          else np.asarray(list(scan_list), dtype=str))
        self.cam_dict = self._cam_dict()
        self.znear, self.zfar = 400 * scale_factor, 1500 * scale_factor
        self.nscans, self.ncams, self.nlights = len(self.scan_list), len(self.cam_dict["ids"]), 7
        self.src_camids = list(SRC_CAM_IDCS)
        self.metas = [dict(scan_idx=s, cam_idx=c, ref_cam_idcs=self.src_camids, light_idx=l)
                      for s, c, l in product(range(self.nscans), range(self.ncams), range(self.nlights))]

    # ---- cameras (dtu.py:157-181): intrinsics x4 (the files describe quarter-resolution images) x downsample, translation x scale
    def _cam_dict(self):
        cam_dir = self.data_dir / "Cameras/train"
        paths = [f for f in sorted(cam_dir.iterdir()) if f.name.endswith("_cam.txt")]
        ids, extr, intr = [], [], []
        for p in paths:
            K, E, _ = formats.read_dtu_cam(str(p))
            K, E = K.copy(), E.copy()
            K[:2] *= 4
            K[:2] = K[:2] * self.downsample
            E[:3, 3] *= self.scale_factor
            ids.append(int(p.name.strip("_cam.txt")))
            extr.append(E)
            intr.append(K)
        return dict(ids=torch.tensor(ids), extrinsics=torch.from_numpy(np.stack(extr)), intrinsics=torch.from_numpy(np.stack(intr)))

    def __len__(self):
        return len(self.metas)

    def read_rgb(self, path):
        """-> (3,h,w) float32 in [0,1], downsampled (dtu.py:71-90)."""
        try:
            from PIL import Image
        except ImportError:
            Image = None
        if Image is not None:
            im = Image.open(path)
            if self.downsample:
                w, h = im.size
                im = im.resize((int(w * self.downsample), int(h * self.downsample)))
            a = np.asarray(im, dtype=np.uint8)
            self._rgb_resample = "PIL.Image.resize (reference path)"
        else:                      # pragma: no cover - PIL is present in the MI355X image
            a = formats.read_png(str(path))
            if self.downsample == 0.5:
                H, W = a.shape[:2]
                a = a[:H // 2 * 2, :W // 2 * 2].reshape(H // 2, 2, W // 2, 2, -1).astype(np.float32).mean((1, 3)).round().astype(np.uint8)
            self._rgb_resample = "2x2 average (PIL absent: NOT the reference's bicubic resampler)"
        if a.ndim == 2:
            a = a[..., None]
        return torch.from_numpy(np.ascontiguousarray(a[..., :3])).permute(2, 0, 1).float() / 255.0

    def read_depth(self, path):
        """TransMVSNet PNG -> depth (1,h,w) in world units + validity mask (dtu.py:92-124)."""
        d = torch.from_numpy(formats.read_transmvsnet_png(str(path), dtu_rescale=True))
        h, w = d.shape
        assert h == 512 and w == 640
        if self.downsample != 1:
            h, w = int(h * self.downsample), int(w * self.downsample)
            d = torch.nn.functional.interpolate(d[None, None], (h, w), mode="nearest")[0, 0]
        mask = (d > 0).float()
        d = d * self.scale_factor
        return d[None], mask[None]

    def read_conf(self, path):
        """confidence PNG (no DTU rescale, but the reference pushes it through read_depth, dtu.py:222-223)."""
        return self.read_depth(path)[0]

    def __getitem__(self, idx):
        m = self.metas[idx]
        scan_name = str(self.scan_list[m["scan_idx"]])
        cam_idcs = [m["cam_idx"]] + list(m["ref_cam_idcs"])
        cam_ids = [self.cam_dict["ids"][i] for i in cam_idcs]
        img_paths = [self.data_dir / "Rectified" / (scan_name + "_train") / f"rect_{int(i) + 1:03d}_{m['light_idx']}_r5000.png"
                     for i in cam_ids]
        depth_paths = [self.data_dir / "Depths" / scan_name / f"depth_map_{int(i):04d}_{self.depth_fname}.png" for i in cam_ids[1:]]
        imgs = torch.stack([self.read_rgb(p) for p in img_paths])
        depths, masks = zip(*[self.read_depth(p) for p in depth_paths])
        stds = torch.stack([self.read_conf(p.parent / p.name.replace(".png", "_conf.png")) for p in depth_paths])
        stds = torch.as_tensor(formats.conf_to_std(stds))
        K = torch.stack([self.cam_dict["intrinsics"][i] for i in cam_idcs])
        E = torch.stack([self.cam_dict["extrinsics"][i] for i in cam_idcs])
        ids = torch.tensor([int(i) for i in cam_ids])
        return dict(target_rgb=imgs[0], target_alpha=torch.ones_like(imgs[0, :1]), target_extrinsics=E[0], target_intrinsics=K[0],
                    target_view_id=ids[0], scan_idx=torch.tensor(m["scan_idx"]), sample_name=f"{scan_name}-{ids[0]}",
                    src_rgbs=imgs[1:], src_alphas=torch.stack(masks), src_depths=torch.stack(depths), src_depth_stds=stds.float(),
                    src_extrinsics=E[1:], src_intrinsics=K[1:], src_view_ids=ids[1:], light_idx=torch.tensor(m["light_idx"]))

    def get_cam_sweep_extrinsics(self, nframes, scan_idx=None, elevation=0.0, radius=0.5):
        """Sweep through cameras 11 -> 24 -> 18 (dtu.py:246-318)."""
        from .sweep import sweep_extrinsics
        E = self.cam_dict["extrinsics"]
        return sweep_extrinsics(E[11], E[24], E[18], nframes)


def collate(samples):
    """list of sample dicts -> batch dict (leading batch dimension on tensors, lists for strings), like the default collate_fn."""
    out = {}
    for k in samples[0]:
        v = [s[k] for s in samples]
        out[k] = torch.stack(v) if torch.is_tensor(v[0]) else v
    return out


def encode_args(batch, device=None):
    """The five arguments of PixelNeRF.encode taken from a batch dict (diner.py:65-70)."""
    mv = (lambda t: t.to(device)) if device is not None else (lambda t: t)
    return dict(images=mv(batch["src_rgbs"]), depths=mv(batch["src_depths"]), depths_std=mv(batch["src_depth_stds"]),
                extrinsics=mv(batch["src_extrinsics"]), intrinsics=mv(batch["src_intrinsics"]))
