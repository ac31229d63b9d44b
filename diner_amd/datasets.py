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

`FacescapeSamples` does the same for the Facescape capture layout (reference src/data/facescape.py): view selection into the
cached sample list ("metas"), the sample dict, the sweep path.

Not built: the Multiface class (same schema, another file tree) and the Lightning DataModule.
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


class FacescapeSamples:
    """Sample dicts from a Facescape tree  root/<subject>/<frame 01..20>/{cameras.json, 3dlmks.npy, view_<id>/{rgba_colorcalib.png,
    depth_TransMVSNet.png, depth_TransMVSNet_conf.png}}  (facescape.py:18-61).

    The sample list is the reference's: per scan, four ideal reference directions at (+-range_hor, +-range_vert) around the frontal
    axis (0, -1, 0), slid in azimuth by multiples of slide_step within +-slide_range; the four cameras nearest to each direction are
    the candidates of that source slot (the first one is used unless random_ref_views); every camera inside the pyramid spanned by the
    four nearest cameras that is not itself one of them is a target (facescape.py:75-207).  The list is cached as JSON next to the
    split files, in the reference's file name and format, and loaded from there when present."""
    znear, zfar = 1.0, 2.5
    RGBA_FNAME = "rgba_colorcalib.png"
    DEPTH_FNAME = "depth_TransMVSNet.png"

    def __init__(self, root, stage, range_hor=45, range_vert=30, slide_range=40, slide_step=20.0, random_ref_views=False,
                 depth_fname=None, split_dir="assets/data_splits/facescape"):
        self.data_dir = Path(root)
        if not self.data_dir.exists():
            raise FileNotFoundError(root)
        self.stage, self.range_hor, self.range_vert = stage, range_hor, range_vert
        self.slide_range, self.slide_step, self.random_ref_views = slide_range, slide_step, random_ref_views
        if depth_fname is not None:
            self.DEPTH_FNAME = depth_fname
        self.DEPTH_STD_FNAME = self.DEPTH_FNAME.replace(".png", "_conf.png")
        self.split_dir = Path(split_dir)
        self.nsource = 4
        self.metas = self._metas()

    @staticmethod
    def conf2std(x):                       # facescape.py:50-52
        return formats.conf_to_std(x, "facescape")

    @staticmethod
    def viewdir(i):
        return f"view_{int(i):05d}"

    @staticmethod
    def read_rgba(path, symmetric_range=False, bg=1.0):
        """-> rgb (3,H,W) in [0,1] (or [-1,1]) with the background (alpha < 0.5) painted `bg`, alpha (1,H,W) (facescape.py:54-62)."""
        a = torch.from_numpy(np.ascontiguousarray(formats.read_png(str(path)))).permute(2, 0, 1).float() / 255.0
        rgb, alpha = a[:3].clone(), a[3:4].clone()
        if symmetric_range:
            rgb = rgb * 2 - 1
        rgb.permute(1, 2, 0)[alpha[0] < 0.5] = bg
        return rgb, alpha

    @staticmethod
    def read_depth(path):
        """uint16 PNG x 1e-4 -> (1,H,W) float32 (facescape.py:64-69)."""
        return torch.from_numpy(formats.read_png(str(path)).astype(np.int32)).float()[None] * 1e-4

    # ---- sample list ------------------------------------------------------------------------------------------------------------
    def _meta_path(self):
        return self.split_dir / (f"{self.stage}_{self.range_hor}_{self.range_vert}" +
                                 (f"_{self.slide_range}" if self.slide_range != 0 else "") + ".txt")

    def _metas(self):
        import json
        mp = self._meta_path()
        if mp.exists():
            with open(mp) as f:
                return json.load(f)
        val_subjects = [f"{int(i):03d}" for i in np.atleast_1d(np.loadtxt(self.split_dir / "publishable_list_v1.txt", delimiter=","))]
        train_subjects = sorted(d.name for d in self.data_dir.iterdir() if d.name not in val_subjects)
        subjects = train_subjects if self.stage == "train" else val_subjects
        metas = []
        for subject, frame in product(subjects, range(1, 21)):
            scan = self.data_dir / subject / f"{frame:02d}"
            try:
                metas += self._scan_metas(scan, first_idx=len(metas))
            except (FileNotFoundError, OSError, KeyError, ValueError):      # the reference skips scans it cannot read
                continue
        with open(mp, "w") as f:
            json.dump(metas, f, indent="\t")
        return metas

    def _scan_metas(self, scan, first_idx):
        import json
        if not (scan / "3dlmks.npy").exists():
            raise FileNotFoundError(scan / "3dlmks.npy")
        with open(scan / "cameras.json") as f:
            cam_dict = json.load(f)

        def usable(i):
            v = scan / self.viewdir(i)
            return ((v / self.RGBA_FNAME).exists() and (v / self.DEPTH_FNAME).exists()
                    and float(self.read_depth(v / self.DEPTH_FNAME).max()) <= self.zfar)
        cam_ids = np.array([i for i in sorted(cam_dict.keys()) if usable(i)])          # ids are strings, sorted as strings
        E = np.array([cam_dict[k]["extrinsics"] for k in cam_ids]).astype(np.float32)
        centre = -E[:, :3, :3].transpose(0, 2, 1) @ E[:, :3, -1:]
        cam_dirs = (centre / np.sqrt((centre ** 2).sum(axis=1, keepdims=True)))[..., 0]
        hor, vert = self.range_hor / 180 * np.pi, self.range_vert / 180 * np.pi
        ideal = np.array([[np.sin(az) * np.cos(el), -np.cos(az) * np.cos(el), np.sin(el)]
                          for az, el in product([-hor, hor], [-vert, vert])])
        # a scan whose frontal view sees nothing nearer than 2 m is dropped
        frontal = cam_ids[np.argmax(np.sum(np.array([0.0, -1.0, 0.0])[None] * cam_dirs, axis=-1))]
        depth = self.read_depth(scan / self.viewdir(frontal) / self.DEPTH_FNAME)
        if depth[depth != 0].min() > 2:
            return []
        out = []
        for slide in np.arange(-self.slide_range, self.slide_range + 1, self.slide_step):
            a = slide / 180 * np.pi
            rot = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0.0, 0.0, 1.0]])
            dirs = (rot @ ideal.T).T
            cos = np.sum(dirs[:, None] * cam_dirs[None], axis=-1)                       # (4, N)
            ref_idcs = np.argsort(cos, axis=1)[:, ::-1][:, :4]
            ref_ids = cam_ids[ref_idcs].tolist()
            corners = cam_dirs[ref_idcs[:, 0]]
            normals = np.stack([np.cross(corners[1], corners[0]), np.cross(corners[3], corners[1]),
                                np.cross(corners[2], corners[3]), np.cross(corners[0], corners[2])], axis=0)
            inside = np.all(np.sum(cam_dirs[:, None] * normals[None], axis=-1) >= 0, axis=-1)
            firsts = [r[0] for r in ref_ids]
            for t in cam_ids[inside].tolist():
                if t in firsts:
                    continue
                out.append(dict(idx=first_idx + len(out), scan_path=str(scan.relative_to(self.data_dir)), target_id=t, ref_ids=ref_ids))
        return out

    def __len__(self):
        return len(self.metas)

    # ---- one sample (facescape.py:217-293) -----------------------------------------------------------------------------------------
    def __getitem__(self, idx):
        import json
        from src.util.cam_geometry import to_homogeneous_trafo
        m = self.metas[idx]
        src_ids = [(np.random.choice(c) if self.random_ref_views else c[0]) for c in m["ref_ids"]]
        tgt = m["target_id"]
        scan = self.data_dir / m["scan_path"]
        frame, subject = scan.name, scan.parent.name
        rgb_t, alpha_t = self.read_rgba(scan / self.viewdir(tgt) / self.RGBA_FNAME)
        rgbs, alphas, depths, stds = [], [], [], []
        for i in src_ids:
            v = scan / self.viewdir(i)
            rgb, alpha = self.read_rgba(v / self.RGBA_FNAME)
            rgbs.append(rgb), alphas.append(alpha)
            depths.append(self.read_depth(v / self.DEPTH_FNAME))
            stds.append(self.read_depth(v / self.DEPTH_STD_FNAME))
        with open(scan / "cameras.json") as f:
            cams = json.load(f)
        E_t = to_homogeneous_trafo(torch.tensor(cams[tgt]["extrinsics"])[None])[0]
        E_s = to_homogeneous_trafo(torch.tensor([cams[i]["extrinsics"] for i in src_ids]))
        return dict(target_rgb=rgb_t, target_alpha=alpha_t, target_extrinsics=E_t,
                    target_intrinsics=torch.tensor(cams[tgt]["intrinsics"]), target_view_id=torch.tensor(int(tgt)), scan_idx=0,
                    sample_name=f"{subject}-{frame}-{tgt}-{'-'.join(src_ids)}-", frame=frame,
                    src_rgbs=torch.stack(rgbs), src_depths=torch.stack(depths), src_depth_stds=self.conf2std(torch.stack(stds)),
                    src_alphas=torch.stack(alphas), src_extrinsics=E_s,
                    src_intrinsics=torch.tensor([cams[i]["intrinsics"] for i in src_ids]),
                    src_view_ids=torch.tensor([int(i) for i in src_ids]))

    def get_cam_sweep_extrinsics(self, nframes, scan_idx, elevation=0.0, radius=1.8, sweep_range=None):
        """Horizontal arc of +-sweep_range degrees around the mean direction of the sample's source cameras, looking at the
        origin with -z up (facescape.py:295-341).  -> (nframes, 4, 4) extrinsics."""
        E = self[scan_idx]["src_extrinsics"]
        centres = -1 * E[:, :3, :3].permute(0, 2, 1) @ E[:, :3, -1:]
        dirs = centres[..., 0] / torch.norm(centres[..., 0], p=2, keepdim=True, dim=-1)
        mean_dir = dirs.sum(dim=0)
        mean_dir = mean_dir / torch.norm(mean_dir, p=2, dim=0)
        centre = mean_dir * radius
        z_ax = -centre / torch.norm(centre, p=2)
        y_ax = torch.tensor([0.0, 0.0, -1.0])
        x_ax = torch.cross(y_ax, z_ax, dim=0)
        x_ax = x_ax / torch.norm(x_ax, p=2)
        pose = torch.eye(4)
        pose[:3, 0], pose[:3, 1], pose[:3, 2], pose[:3, 3] = x_ax, y_ax, z_ax, centre
        rng = sweep_range if sweep_range is not None else self.range_hor
        rots = torch.stack([torch.tensor([[np.cos(a), -np.sin(a), 0, 0], [np.sin(a), np.cos(a), 0, 0], [0.0, 0.0, 1, 0],
                                          [0.0, 0.0, 0.0, 1.0]], dtype=torch.float)
                            for a in np.linspace(-rng / 180 * np.pi, rng / 180 * np.pi, nframes)])
        return torch.linalg.inv(rots @ pose[None].expand(nframes, -1, -1))


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
