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


def oracle_setup(W, H, seed, bg_std_zero=False, **scene_kw):
    """-> (scene dict, oracle Scene, oracle MLPWeights, mlp state_dict, target rays (H*W,8))"""
    sc = make_scene(W, H, seed=seed, bg_std_zero=bg_std_zero, **scene_kw)
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


SAT_L = 2.5e-7   # absolute likelihood uncertainty between two erf implementations, see selection_diff


def selection_diff(ref_u, got_u, L, zc, n_pick, skip=()):
    """Rays whose depth-guided picks differ between two implementations, and how far (in likelihood) the candidates that
    are in one pick set but not in the other lie from the ray's cut-off likelihood.

    The sampler keeps the n_pick = K - G candidates of largest likelihood L = 0.5 * |erf(a) - erf(b)| that are > 0
    (nerf_renderer.py:172-178).  Away from the surface both erf values are within a few ulp (6e-8) of +-1, so L is a small
    multiple of 3e-8 whose last step depends on the erf implementation (Sleef's AVX-512 kernel on the Intel host that
    generated the fixtures, another Sleef kernel on an AMD host, ocml on the GPU, CUDA's erff for the reference on an A100
    -- they all differ by an ulp here and there).  One erf ulp moves L by up to 6e-8; two implementations can therefore
    disagree about the ORDER of two candidates whose likelihoods are within ~2.4e-7 of each other, and about whether a
    candidate with L <= 1.2e-7 counts as "L > 0".  When such candidates straddle the cut-off (the n_pick-th largest L of
    the ray, or 0 when fewer are positive) the pick is implementation-defined in the reference itself.  Everything
    further than SAT_L from the cut-off must match exactly.  ref_u / got_u: per-ray ascending unfilled z; `skip`: rays
    with an exact tie at the cut-off (the reference's pick there is its unstable sort's)."""
    bad = (~torch.isclose(got_u, ref_u, rtol=3e-6, atol=1e-7).all(-1)).nonzero().flatten().tolist()
    worst = 0.0
    for r in bad:
        if r in skip:
            continue
        Ls = L[r].sort(descending=True).values
        cut = float(Ls[n_pick - 1]) if n_pick <= Ls.numel() else 0.0
        only = set(ref_u[r].tolist()) ^ set(got_u[r].tolist())
        for zz in only:
            i = (zc[r] == zz).nonzero().flatten()
            if len(i):                       # a candidate depth (not a gaussian / fill sample)
                worst = max(worst, abs(float(L[r, i[0]]) - cut))
    return bad, worst
