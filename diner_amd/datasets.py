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

`MultifaceSamples` does it for the Multiface layout (reference src/data/multiface.py): KRT camera files, gamma-corrected images,
separate mask and depth trees, millimetre extrinsics.

Not built: augmentation and the Lightning DataModule.
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


class MultifaceSamples:
    """Sample dicts from a Multiface tree  root/<subject>/{KRT, images/<seq>/<cam>/<frame>.png, masks/<seq>/<cam>/<frame>.png,
    depths/<seq>/<cam>/<frame><suffix>}  (multiface.py:22-381).

    Sample list: per subject of the split file, the camera nearest to each of the four ideal reference centres is a source view; every
    other camera within 100 mm of the inner side of the four planes through neighbouring reference cameras is a target; one entry per
    (sequence, target camera, frame).  Cached as JSON next to the split files under the reference's name, optional substring filters
    applied after loading (multiface.py:133-247)."""
    znear, zfar = 0.5, 1.5

    def __init__(self, root, stage, downsample=8, split_config="assets/data_splits/multiface/tiny_subset.json", depth_suffix=".png",
                 depth_std_suffix=None, subject_filter=None, sequence_filter=None, target_filter=None, manual_target_params=None,
                 split_dir="assets/data_splits/multiface"):
        import json
        self.data_dir = Path(root)
        if not self.data_dir.exists():
            raise FileNotFoundError(root)
        assert isinstance(downsample, int)
        self.stage, self.downsample, self.nsource = stage, downsample, 4
        self.split_config, self.split_dir = Path(split_config), Path(split_dir)
        self.depth_suffix, self.depth_std_suffix = depth_suffix, depth_std_suffix
        self.metas = self._metas(subject_filter, sequence_filter, target_filter)
        self.manual_target_params = None
        if manual_target_params is not None:
            with open(manual_target_params) as f:
                self.manual_target_params = json.load(f)
            assert len(self.manual_target_params["extrinsics"]) == len(self)

    # ---- file readers -------------------------------------------------------------------------------------------------------------
    @staticmethod
    def gamma_correct(img, dim=0):
        """The capture's colour pipeline: per-channel gains (1.4, 1.1, 1.6) / 1.1, black level 3/255, gamma 2 (multiface.py:79-99)."""
        gamma, black = 2.0, 3.0 / 255.0
        scale = torch.tensor([1.4, 1.1, 1.6]).view([3 if i == dim else 1 for i in range(img.dim())])
        img = img * scale.to(img) / 1.1
        return torch.clamp((((1.0 / (1 - black)) * 0.95 * torch.clamp(img - black, 0, 2)) ** (1.0 / gamma)) - 15.0 / 255.0, 0, 2)

    @classmethod
    def read_img(cls, path, symmetric_range=False):
        a = formats.read_png(str(path))
        rgb = torch.from_numpy(np.ascontiguousarray(a)).permute(2, 0, 1).float() / 255.0
        rgb = cls.gamma_correct(rgb, dim=0).clip(0, 1)
        return rgb * 2 - 1 if symmetric_range else rgb

    @staticmethod
    def read_alpha(path):
        a = formats.read_png(str(path))
        a = a[..., None] if a.ndim == 2 else a
        return torch.from_numpy(np.ascontiguousarray(a)).permute(2, 0, 1).float() / 255.0

    @staticmethod
    def read_depth(path):
        return torch.from_numpy(formats.read_png(str(path)).astype(np.int32)).float()[None] * 1e-4

    @staticmethod
    def load_krt(path):
        """KRT file: per camera a name line, 3 intrinsics rows, a distortion row, 3 extrinsics rows (3x4, mm), a blank line."""
        cams = {}
        with open(path) as f:
            while True:
                name = f.readline()
                if name == "":
                    break
                intrin = [[float(x) for x in f.readline().split()] for _ in range(3)]
                dist = [float(x) for x in f.readline().split()]
                extrin = [[float(x) for x in f.readline().split()] for _ in range(3)]
                f.readline()
                cams[name[:-1]] = dict(intrin=np.array(intrin), dist=np.array(dist), extrin=np.array(extrin))
        return cams

    # ---- sample list --------------------------------------------------------------------------------------------------------------
    def _metas(self, subject_filter, sequence_filter, target_filter):
        import json
        mp = self.split_dir / f"{self.stage}_{self.split_config.stem}.txt"
        if mp.exists():
            with open(mp) as f:
                metas = json.load(f)
        else:
            with open(self.split_config) as f:
                cfg = json.load(f)["train" if self.stage == "train" else "val"]
            metas = []
            for subj in cfg["subjects"]:
                krt = self.load_krt(self.data_dir / subj / "KRT")
                names = np.array(sorted(krt.keys()))
                E = np.array([krt[n]["extrin"] for n in names])
                E = np.concatenate((E, np.zeros_like(E[:, :1])), axis=1)
                E[:, -1, -1] = 1
                centres = (-E[:, :3, :3].transpose(0, 2, 1) @ E[:, :3, -1:])[..., 0]
                dirs = E[:, 2, :3]
                origin = np.array([[0, 0, 1000.0]])
                ideal = np.array(cfg["ref_centers"]).reshape(-1, 3)
                if subj == "m--20190529--1004--5067077--GHS":          # this capture's rig is rotated (multiface.py:161-166)
                    b = np.pi * 4 / 6
                    rot_y = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
                    ideal = (rot_y @ (ideal - origin).T).T + origin
                dist = np.sqrt(np.sum((ideal[:, None] - centres[None]) ** 2, axis=-1))
                ref = np.argsort(dist, axis=1)[:, 0]
                rc, rd = centres[ref], dirs[ref]
                normals = np.cross(rc[[0, 1, 2, 3]] - rc[[1, 2, 3, 0]], rd[[0, 1, 2, 3]] + rd[[1, 2, 3, 0]])
                normals = normals / np.sqrt(np.sum(normals ** 2, axis=-1, keepdims=True))
                inside = np.all(np.sum((centres[None] - rc[:, None]) * normals[:, None], axis=-1) > -100, axis=0)
                inside[ref] = False
                targets, ref_names = names[inside].tolist(), names[ref].tolist()
                for seq in [p for p in sorted((self.data_dir / subj / "images").iterdir()) if p.name in cfg["sequences"]]:
                    for t in targets:
                        for frame in sorted((seq / t).iterdir()):
                            metas.append(dict(idx=len(metas), scan_path=str(frame.relative_to(self.data_dir)), target_id=t,
                                              ref_ids=ref_names))
            with open(mp, "w") as f:
                json.dump(metas, f, indent="\t")
        if subject_filter is not None:
            metas = [m for m in metas if any(s in m["scan_path"] for s in subject_filter)]
        if sequence_filter is not None:
            metas = [m for m in metas if any(s in m["scan_path"] for s in sequence_filter)]
        if target_filter is not None:
            metas = [m for m in metas if any(t == m["target_id"] for t in target_filter)]
        return metas

    def __len__(self):
        return len(self.metas)

    def _dpath(self, p, suffix):
        return p.parents[3] / "depths" / p.relative_to(p.parents[2]).parent / (p.stem + suffix)

    @staticmethod
    def _apath(p):
        return p.parents[3] / "masks" / p.relative_to(p.parents[2])

    # ---- one sample (multiface.py:267-381) --------------------------------------------------------------------------------------------
    def __getitem__(self, idx):
        from src.util.cam_geometry import to_homogeneous_trafo
        m = self.metas[idx]
        src_ids, tgt = m["ref_ids"], m["target_id"]
        rel = Path(m["scan_path"])
        subject, seq, frame = rel.parents[3].name, rel.parents[1].name, rel.stem
        tpath = self.data_dir / rel
        spaths = [self.data_dir / subject / "images" / seq / s / (frame + ".png") for s in src_ids]
        rgb_t, alpha_t = self.read_img(tpath), self.read_alpha(self._apath(tpath))
        rgbs, alphas, depths, stds = [], [], [], []
        for p in spaths:
            d = self.read_depth(self._dpath(p, self.depth_suffix))
            if self.depth_std_suffix is None:
                std = torch.ones_like(d) * 1e-3
                std[d == 0] = 0
            else:
                std = formats.conf_to_std(self.read_depth(self._dpath(p, self.depth_std_suffix)), "multiface", depth=d)
            rgbs.append(self.read_img(p)), alphas.append(self.read_alpha(self._apath(p))), depths.append(d), stds.append(std)
        rgbs, alphas, depths, stds = torch.stack(rgbs), torch.stack(alphas), torch.stack(depths), torch.stack(stds)
        rgbs.permute(0, 2, 3, 1)[alphas[:, 0] < 1] = 1                       # white background
        rgb_t.permute(1, 2, 0)[alpha_t[0] < 1] = 1
        cams = self.load_krt(self.data_dir / subject / "KRT")
        if self.manual_target_params is None:
            E_t, K_t = torch.tensor(cams[tgt]["extrin"]).float(), torch.tensor(cams[tgt]["intrin"]).float()
        else:
            E_t = torch.tensor(self.manual_target_params["extrinsics"][idx]).float()
            K_t = torch.tensor(self.manual_target_params["intrinsics"][idx]).float()
        E_s = torch.tensor(np.array([cams[s]["extrin"] for s in src_ids])).float()
        E_t, E_s = to_homogeneous_trafo(E_t[None]).float()[0], to_homogeneous_trafo(E_s).float()
        K_s = torch.tensor(np.array([cams[s]["intrin"] for s in src_ids])).float()
        E_t[..., :3, -1] /= 1000                                             # mm -> m
        E_s[..., :3, -1] /= 1000
        H, W = rgb_t.shape[-2:]
        h, w = int((H / self.downsample) // 32 * 32), int((W / self.downsample) // 32 * 32)
        if h != H or w != W:
            # torchvision.transforms.functional.resize on tensors: bilinear without antialiasing for images, nearest for masks / depths
            F = torch.nn.functional

            def bil(x):
                return F.interpolate(x if x.dim() == 4 else x[None], (h, w), mode="bilinear", align_corners=False)[slice(None) if x.dim() == 4 else 0]

            def nn(x):
                return F.interpolate(x if x.dim() == 4 else x[None], (h, w), mode="nearest")[slice(None) if x.dim() == 4 else 0]
            rgb_t, rgbs, alpha_t, alphas = bil(rgb_t), bil(rgbs), nn(alpha_t), nn(alphas)
            if depths.shape[-2:] != rgbs.shape[-2:]:
                depths, stds = nn(depths), nn(stds)
            K_t[0] *= w / W
            K_t[1] *= h / H
            K_s[:, 0] *= w / W
            K_s[:, 1] *= h / H
        return dict(target_rgb=rgb_t, target_alpha=alpha_t, target_extrinsics=E_t, target_intrinsics=K_t,
                    target_view_id=torch.tensor(int(tgt)), scan_idx=0, sample_name=f"{subject}-{seq}-{frame}-{tgt}-{'-'.join(src_ids)}",
                    frame=frame, src_rgbs=rgbs, src_depths=depths, src_depth_stds=stds, src_alphas=alphas, src_extrinsics=E_s,
                    src_intrinsics=K_s, src_view_ids=torch.tensor([int(s) for s in src_ids]))

    def get_cam_sweep_extrinsics(self, nframes, scan_idx, elevation=0.0, radius=1.8, sweep_range=None):
        """Closed loop through the four source cameras (0, 1, 2, 3, 0, 2): spherical interpolation of the orientations, piece-wise
        linear interpolation of the positions (multiface.py:383-430)."""
        from scipy.spatial.transform import Rotation
        from src.util.cam_geometry import Slerp
        pose = torch.linalg.inv(self[scan_idx]["src_extrinsics"])
        rots = Rotation.from_matrix(pose[:, :3, :3].cpu().numpy())
        rots = Rotation.concatenate((rots, rots[0], rots[2]))
        c = pose[:, :3, -1]
        c = torch.cat((c, c[0][None], c[2][None]), dim=0).cpu().numpy()
        slerp = Slerp(np.linspace(0, 1, len(rots)), rots, c)
        R, t = slerp(np.linspace(0, 1, nframes + 1)[:-1])
        poses = np.repeat(np.eye(4)[None], nframes, axis=0)
        poses[:, :3, :3] = R.as_matrix()
        poses[:, :3, -1] = t
        return torch.linalg.inv(torch.from_numpy(poses)).float()


def collate(samples):
    """list of sample dicts -> batch dict (leading batch dimension on tensors, lists for strings), like the default collate_fn."""
    out = {}
    for k in samples[0]:
        v = [s[k] for s in samples]
        if torch.is_tensor(v[0]):
            out[k] = torch.stack(v)
        elif isinstance(v[0], (bool, int, float)):         # numeric scalars become tensors, as torch's default_collate does
            out[k] = torch.tensor(v, dtype=torch.bool if isinstance(v[0], bool) else torch.float64 if isinstance(v[0], float) else torch.int64)
        else:
            out[k] = v
    return out


def encode_args(batch, device=None):
    """The five arguments of PixelNeRF.encode taken from a batch dict (diner.py:65-70)."""
    mv = (lambda t: t.to(device)) if device is not None else (lambda t: t)
    return dict(images=mv(batch["src_rgbs"]), depths=mv(batch["src_depths"]), depths_std=mv(batch["src_depth_stds"]),
                extrinsics=mv(batch["src_extrinsics"]), intrinsics=mv(batch["src_intrinsics"]))
