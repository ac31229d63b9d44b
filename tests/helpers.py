"""Shared test plumbing: rebuild the seeded synthetic inputs the golden vectors were generated from."""
import hashlib
import os

import numpy as np
import torch

from diner_amd.synthetic import make_scene, make_mlp_state_dict
from src.util.depth2normal import depth2normal
from oracle import diner_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def sha(*tensors):
    h = hashlib.sha256()
    for t in tensors:
        h.update(np.ascontiguousarray(t.detach().cpu().numpy()).tobytes())
    return h.hexdigest()


def load(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)


def oracle_setup(W, H, seed, bg_std_zero=False):
    """-> (scene dict, oracle Scene, oracle MLPWeights, mlp state_dict, target rays (H*W,8))"""
    sc = make_scene(W, H, seed=seed, bg_std_zero=bg_std_zero)
    normals = depth2normal(sc["depths"], sc["src_intrinsics"])
    sc["normals"] = normals
    K = sc["src_intrinsics"]
    scene = O.Scene(latent=sc["latent"], depths=sc["depths"], depths_std=sc["depths_std"], normals=normals,
                    poses=sc["src_extrinsics"], focal=K[:, [0, 1], [0, 1]], c=K[:, :2, -1],
                    image_shape=sc["image_shape"], feature_padding=sc["feature_padding"])
    msd = make_mlp_state_dict()
    w = O.MLPWeights.from_state_dict(msd)
    rays = O.gen_rays(sc["target_extrinsics"], sc["target_intrinsics"], W, H, sc["znear"], sc["zfar"])
    return sc, scene, w, msd, rays


def max_norm_rel(a, b):
    a, b = torch.as_tensor(a).float().cpu(), torch.as_tensor(b).float().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp(min=1e-30)).item()
