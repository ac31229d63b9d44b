"""CPU oracle for DINER's volumetric-rendering hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a plain PyTorch-CPU (fp32) restatement of the reference algorithm
(malteprinzler/diner, mounted read-only at /root/reference during the build).  It is
the checker for the HIP kernels, never the thing measured or shipped: only tests/,
__graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import it.  The product
path (diner_amd/, src/) never imports anything from oracle/.

Pinning: the reference repository has no tests or golden vectors for this path
(SURVEY.md section 4), and the arithmetic lives in torch.  The oracle is therefore pinned
against the *imported reference modules themselves*, run in the build container by
oracle/make_golden.py; the resulting input/output vectors are committed under
tests/golden/ and checked by tests/test_oracle_golden.py on every run (CPU).

All random draws of the reference (nerf_renderer.py:57, :188, :390) are replaced by
explicit full-shape noise tensors so that results are reproducible:
    noise_coarse (NR, n_cand)  uniform [0,1)   -- stratified candidate jitter
    noise_gauss  (NR, G)       standard normal -- gaussian depth samples
    noise_fill   (NR, K)       uniform [0,1)   -- stratified fill, indexed by SORTED column

Conventions: one object (the reference's SB dim is dropped),
NV source views, extrinsics world->camera (OpenCV), image_shape = [W, H].
"""
import math
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------
# containers
# --------------------------------------------------------------------------------------
@dataclass
class Scene:
    """Per-object state the reference keeps on PixelNeRF / SpatialEncoder after encode()
    (pixelnerf.py:47-51, image_encoder.py:232-236, :290-291)."""
    latent: torch.Tensor      # (NV, C, Hf, Wf)
    depths: torch.Tensor      # (NV, 1, Hs, Ws)
    depths_std: torch.Tensor  # (NV, 1, Hs, Ws)
    normals: torch.Tensor     # (NV, 3, Hs, Ws)
    poses: torch.Tensor       # (NV, 4, 4) world->cam
    focal: torch.Tensor       # (NV, 2)
    c: torch.Tensor           # (NV, 2)
    image_shape: torch.Tensor  # (2,) = [W, H] as float
    feature_padding: float = 32.0

    @property
    def nv(self):
        return self.poses.shape[0]


@dataclass
class MLPWeights:
    """ResnetFC parameters (resnetfc.py:72-127), nn.Linear layout (out, in)."""
    lin_in_w: torch.Tensor
    lin_in_b: torch.Tensor
    lin_z_w: list
    lin_z_b: list
    fc0_w: list
    fc0_b: list
    fc1_w: list
    fc1_b: list
    lin_out_w: torch.Tensor
    lin_out_b: torch.Tensor
    combine_layer: int = 3
    d_latent: int = 512

    @staticmethod
    def from_state_dict(sd, prefix="", combine_layer=3, d_latent=512):
        g = lambda k: sd[prefix + k].detach().float().cpu().contiguous()
        n_blocks = len([k for k in sd if k.startswith(prefix + "blocks.") and k.endswith("fc_0.weight")])
        n_z = len([k for k in sd if k.startswith(prefix + "lin_z.") and k.endswith(".weight")])
        return MLPWeights(
            lin_in_w=g("lin_in.weight"), lin_in_b=g("lin_in.bias"),
            lin_z_w=[g(f"lin_z.{i}.weight") for i in range(n_z)],
            lin_z_b=[g(f"lin_z.{i}.bias") for i in range(n_z)],
            fc0_w=[g(f"blocks.{i}.fc_0.weight") for i in range(n_blocks)],
            fc0_b=[g(f"blocks.{i}.fc_0.bias") for i in range(n_blocks)],
            fc1_w=[g(f"blocks.{i}.fc_1.weight") for i in range(n_blocks)],
            fc1_b=[g(f"blocks.{i}.fc_1.bias") for i in range(n_blocks)],
            lin_out_w=g("lin_out.weight"), lin_out_b=g("lin_out.bias"),
            combine_layer=combine_layer, d_latent=d_latent)


# --------------------------------------------------------------------------------------
# a6  positional encoding            (positional_encoding.py:14-53)
# --------------------------------------------------------------------------------------
def posenc(x, num_freqs=6, freq_factor=6.28, include_input=True):
    """x (..., d) -> (..., d*(2F [+1])).  Output order: [x, then for j in 0..2F-1 (freq j//2,
    phase 0 | fp32(pi/2)) for d: sin(phase_j + x_d * freq_j)]  (positional_encoding.py:45-49)."""
    shp = x.shape
    d = shp[-1]
    xf = x.reshape(-1, d)
    freqs = freq_factor * 2.0 ** torch.arange(0, num_freqs)             # :18 (float32)
    f2 = torch.repeat_interleave(freqs, 2).view(1, -1, 1)               # :25
    ph = torch.zeros(2 * num_freqs)
    ph[1::2] = np.pi * 0.5                                              # :30
    ph = ph.view(1, -1, 1)
    rep = xf.unsqueeze(1).repeat(1, 2 * num_freqs, 1)
    emb = torch.sin(torch.addcmul(ph, rep, f2)).view(xf.shape[0], -1)   # :46
    if include_input:
        emb = torch.cat((xf, emb), dim=-1)
    return emb.reshape(*shp[:-1], emb.shape[-1])


# --------------------------------------------------------------------------------------
# a8  the four gathers               (image_encoder.py:97-223, torch_helpers.py:99-159)
# --------------------------------------------------------------------------------------
def _gs(img, uv, mode, padding):
    """img (NV,C,H,W), uv (NV,N,2) -> (NV,C,N) through ATen grid_sample, align_corners=False."""
    return F.grid_sample(img, uv.unsqueeze(2), mode=mode, padding_mode=padding, align_corners=False)[..., 0]


def index_latent(scene: Scene, uv):
    """bilinear / border on the padded feature map after shrinking uv by the feature padding
    (image_encoder.py:112-123)."""
    size = torch.tensor([scene.latent.shape[-1], scene.latent.shape[-2]])
    uv = uv * ((size - scene.feature_padding * 2) / size).view(1, 1, 2)
    return _gs(scene.latent, uv, "bilinear", "border")


def index_depth(scene: Scene, uv):
    """nearest / border (image_encoder.py:157-167)."""
    return _gs(scene.depths, uv, "nearest", "border")


def exponential_padding(img, pad, double_width):
    """replicate-pad by `pad` px, value scaled by 2^(e/double_width) with e = ring index - 1
    clipped at 0 (Chebyshev ring); evaluated as base*exp(e/double_width*ln2) in fp32
    (torch_helpers.py:110-120)."""
    N, C, H, W = img.shape
    base = F.pad(img, [pad] * 4, mode="replicate")
    ry = torch.arange(H + 2 * pad)
    rx = torch.arange(W + 2 * pad)
    ky = torch.clamp(torch.maximum(pad - ry, ry - (H + pad - 1)), min=0)     # 0 inside, 1 = first ring
    kx = torch.clamp(torch.maximum(pad - rx, rx - (W + pad - 1)), min=0)
    e = torch.clamp(torch.maximum(ky.view(-1, 1), kx.view(1, -1)) - 1, min=0).to(img.dtype)
    e = e.view(1, 1, H + 2 * pad, W + 2 * pad).expand(N, C, -1, -1)
    return base * torch.exp(e / double_width * np.log(2))


def index_depth_std(scene: Scene, uv, pad=100, double_width=12):
    """nearest on the exponentially padded std map, zeros outside the pad ring
    (image_encoder.py:185-194, torch_helpers.py:149-159)."""
    H, W = scene.depths_std.shape[-2:]
    size = torch.tensor([W, H], dtype=torch.float)
    padded = exponential_padding(scene.depths_std, pad, double_width)
    uv = uv * (size / (size + 2 * pad)).view(1, 1, 2)
    return _gs(padded, uv, "nearest", "zeros")


def index_normal(scene: Scene, uv):
    """nearest / zeros (image_encoder.py:210-220)."""
    return _gs(scene.normals, uv, "nearest", "zeros")


# --------------------------------------------------------------------------------------
# shared geometry
# --------------------------------------------------------------------------------------
def _fma32(a, b, c):
    """fp32 fused multiply-add, emulated through float64 (a*b is exact there)."""
    return (a.double() * b.double() + c.double()).float()


def rot3(R, x):
    """R (NV,3,3) applied to x (B,3) or (NV,B,3) -> (NV,B,3).

    The reference writes this as torch.matmul(R, x^T)^T (pixelnerf.py:92,:100; nerf_renderer.py:100,:103).
    MKL's sgemm evaluates the K=3 contraction as r0*x0, fma(r1,x1,.), fma(r2,x2,.) on the machine the oracle
    was pinned on, but its kernel choice (hence the last bit of the result) differs between CPUs; nearest-
    neighbour taps downstream turn such a bit into a different pixel.  The oracle therefore spells the
    contraction out -- bit-identical to the reference's matmul where it was pinned (oracle/make_golden.py
    asserts this on every fixture) and identical on every host."""
    if x.dim() == 2:
        x = x.unsqueeze(0)
    x0, x1, x2 = x[..., 0], x[..., 1], x[..., 2]
    rows = []
    for i in range(3):
        r0, r1, r2 = (R[:, i, k].view(-1, 1) for k in range(3))
        rows.append(_fma32(r2, x2, _fma32(r1, x1, r0 * x0)))
    return torch.stack(rows, dim=-1)


def world_to_cam(scene: Scene, xyz):
    """xyz (B,3) world -> (NV,B,3) camera frames: R x + t  (pixelnerf.py:91-93, nerf_renderer.py:99-101)."""
    return rot3(scene.poses[:, :3, :3], xyz) + scene.poses[:, :3, -1].unsqueeze(-2)


def project_uv(scene: Scene, xyz_cam):
    """(NV,B,3) -> normalised uv in [-1,1], outer pixel edges at +-1 (pixelnerf.py:105-108)."""
    uv = xyz_cam[..., :2] / xyz_cam[..., 2:]
    uv = uv * scene.focal.unsqueeze(-2)
    uv = uv + scene.c.unsqueeze(-2)
    return uv / scene.image_shape * 2 - 1


# --------------------------------------------------------------------------------------
# a1  stratified candidates          (nerf_renderer.py:39-63)
# --------------------------------------------------------------------------------------
def sample_coarse(rays, n_cand, noise_coarse):
    near, far = rays[:, 6:7], rays[:, 7:8]
    step = 1.0 / n_cand
    t = torch.linspace(0, 1 - step, n_cand).unsqueeze(0).repeat(rays.shape[0], 1)
    t = t + noise_coarse * step
    return near * (1 - t) + far * t


# --------------------------------------------------------------------------------------
# a2  depth-guided likelihoods + selection + gaussian samples   (nerf_renderer.py:65-190)
# --------------------------------------------------------------------------------------
def point_likelihood(scene: Scene, rays, z_cand, depth_diff_max=0.05):
    """Returns (pt_likelihood, opaque_likelihood), both (NR, n_cand)."""
    NR, n_cand = z_cand.shape
    NV = scene.nv
    step_size = (rays[:, 7] - rays[:, 6]) / n_cand                               # :95
    xyz = rays[:, None, :3] + z_cand.unsqueeze(-1) * rays[:, None, 3:6]           # :96
    xyz_cam = world_to_cam(scene, xyz.reshape(-1, 3))                            # :99-101
    dirs_cam = rot3(scene.poses[:, :3, :3], rays[:, 3:6])                         # :103
    pdirs_cam = dirs_cam.repeat_interleave(n_cand, dim=-2)                       # :104
    uv = project_uv(scene, xyz_cam)                                              # :107-110
    ref_d = index_depth(scene, uv)                                               # (NV,1,B)
    ref_s = index_depth_std(scene, uv)
    ref_n = index_normal(scene, uv)                                              # (NV,3,B)
    ref_z = xyz_cam[..., 2:].permute(0, 2, 1)                                    # (NV,1,B)
    ss = step_size.repeat_interleave(n_cand).view(1, 1, -1).expand_as(ref_d)
    pd = pdirs_cam.transpose(-2, -1)                                             # (NV,3,B)
    cosd = ((pd[:, 0:1] * ref_n[:, 0:1] + pd[:, 1:2] * ref_n[:, 1:2]) + pd[:, 2:3] * ref_n[:, 2:3])   # :119 (sum over 3)
    mask = (ref_s != 0) & ((ref_d - ref_z).abs() < depth_diff_max) & (cosd <= 0)  # :121-124
    L = torch.zeros_like(ref_d)
    sq2 = np.sqrt(2)
    L[mask] = 0.5 * (torch.special.erf((ref_z[mask] + ss[mask] / 2 - ref_d[mask]) / (ref_s[mask] * sq2))
                     - torch.special.erf((ref_z[mask] - ss[mask] / 2 - ref_d[mask]) / (ref_s[mask] * sq2))).abs()
    L = torch.max(L, dim=0).values.squeeze(0).reshape(NR, n_cand)                 # :129-130
    O = L.clone()
    O[:, 1:] *= torch.cumprod(1.0 - L, dim=-1)[:, :-1]                            # :131-132
    return L, O


def weighted_mean_n_std(x, w):
    """torch_helpers.py:215-223 with dim=-1, keepdims=True."""
    wn = w / w.sum(dim=-1, keepdim=True)
    mean = (x * wn).sum(dim=-1, keepdim=True)
    std = ((x - mean).pow(2) * wn).sum(dim=-1, keepdim=True).sqrt()
    return mean, std


def sample_depthguided(scene: Scene, rays, n_samples, n_cand, n_gaussian, noise_coarse, noise_gauss,
                       depth_diff_max=0.05, return_aux=False):
    """(NR, K) unsorted z with exact zeros marking empty slots (nerf_renderer.py:94-190)."""
    assert n_samples >= n_gaussian
    NR = rays.shape[0]
    z_cand = sample_coarse(rays, n_cand, noise_coarse)
    L, O = point_likelihood(scene, rays, z_cand, depth_diff_max)
    idx = L.argsort(dim=-1, descending=True)[:, :n_samples]                       # :172
    Lsel = torch.gather(L, 1, idx)
    z = torch.gather(z_cand, 1, idx)
    z[Lsel == 0.] = 0                                                            # :176-178
    if n_gaussian > 0:
        ray_mask = torch.any(O != 0, dim=-1)                                      # :182
        g = torch.zeros(NR, n_gaussian)
        if ray_mask.any():
            mu, sd = weighted_mean_n_std(z_cand[ray_mask], O[ray_mask])
            g[ray_mask] = noise_gauss[ray_mask] * sd + mu                          # :188
        z[:, -n_gaussian:] = g                                                    # :190
    if return_aux:
        return z, dict(z_cand=z_cand, L=L, O=O)
    return z


# --------------------------------------------------------------------------------------
# a4  stratified fill of the empty slots   (nerf_renderer.py:367-397)
# --------------------------------------------------------------------------------------
def fill_up_uniform_samples(z, rays, noise_fill):
    z = z.sort(dim=-1).values.clone()                                            # :377
    miss = z == 0
    iray, isamp = torch.where(miss)                                              # :381
    n_missing = miss.int().sum(dim=-1)[iray]
    near, far = rays[iray, 6], rays[iray, 7]
    step = (far - near) / n_missing                                              # :388
    zm = near + isamp * step
    zm = zm + noise_fill[iray, isamp] * step                                     # :390
    z[iray, isamp] = zm
    return z.sort(dim=-1).values                                                 # :396


# --------------------------------------------------------------------------------------
# a7  ResnetFC                        (resnetfc.py:61-69, :129-159)
# --------------------------------------------------------------------------------------
def mlp_forward(w: MLPWeights, zx, relu_masks=None):
    """zx (NV, B, d_latent + d_in) -> (B, d_out); views averaged before block `combine_layer`.

    relu_masks (test aid, NOT reference behaviour): dict(X=[5 bool tensors], H=[5], last=bool tensor) -- every relu(t) of
    resnetfc.py:61-69 / :159 is evaluated as t * mask with the decisions of ANOTHER evaluation of the same network (the HIP
    training forward's saved pre-activations).  Two fp32 evaluations put a few pre-activations that are within rounding of zero on
    different sides of the relu; conditioned on one set of decisions, the gradients of the two must agree to round-off."""
    act = (lambda t, m: torch.relu(t)) if relu_masks is None else (lambda t, m: t * m.to(t.dtype))
    mk = (lambda k, b=None: None) if relu_masks is None else (lambda k, b=None: relu_masks[k] if b is None else relu_masks[k][b])
    z = zx[..., :w.d_latent]
    x = F.linear(zx[..., w.d_latent:], w.lin_in_w, w.lin_in_b)
    for b in range(len(w.fc0_w)):
        if b == w.combine_layer:
            x = torch.mean(x, dim=0)
        if b < w.combine_layer:
            x = x + F.linear(z, w.lin_z_w[b], w.lin_z_b[b])
        net = F.linear(act(x, mk("X", b)), w.fc0_w[b], w.fc0_b[b])
        x = x + F.linear(act(net, mk("H", b)), w.fc1_w[b], w.fc1_b[b])
    return F.linear(act(x, mk("last")), w.lin_out_w, w.lin_out_b)


# --------------------------------------------------------------------------------------
# a5  PixelNeRF.forward               (pixelnerf.py:55-145)
# --------------------------------------------------------------------------------------
def mlp_input(scene: Scene, xyz, viewdirs, num_freqs=6, freq_factor=6.28):
    """(B,3),(B,3) -> (NV,B,d_latent+55): [latent | x_c, 36 sincos | R d | dd, 12 sincos]."""
    NV = scene.nv
    xc = world_to_cam(scene, xyz)                                                # :91-93
    zf = posenc(xc, num_freqs, freq_factor)                                      # :96
    vd = rot3(scene.poses[:, :3, :3], viewdirs)                                  # :100
    zf = torch.cat((zf, vd), dim=-1)
    uv = project_uv(scene, xc)
    lat = index_latent(scene, uv).transpose(-1, -2)                              # (NV,B,C)
    dd = index_depth(scene, uv).squeeze(-2) - xc[..., -1]                        # :114-115
    df = posenc(dd.unsqueeze(-1), num_freqs, freq_factor)
    return torch.cat((lat, zf, df), dim=-1)                                      # :128


def pixelnerf_forward(scene: Scene, w: MLPWeights, xyz, viewdirs, relu_masks=None):
    out = mlp_forward(w, mlp_input(scene, xyz, viewdirs), relu_masks)
    sigma = torch.relu(out[..., 3:4]) if relu_masks is None else out[..., 3:4] * relu_masks["sigma"].to(out.dtype)   # (test aid, see mlp_forward)
    return torch.cat([torch.sigmoid(out[..., :3]), sigma], dim=-1)                       # :139-143


# --------------------------------------------------------------------------------------
# a9  compositor                      (nerf_renderer.py:286-365)
# --------------------------------------------------------------------------------------
def composite_from_field(sigma_rgb, rays, z, white_bkgd):
    """sigma_rgb (NR,K,4) = [r,g,b,sigma]; returns weights (NR,K), rgb (NR,3), depth (NR)."""
    deltas = torch.cat([z[:, 1:] - z[:, :-1], rays[:, 7:8] - z[:, -1:]], -1)      # :299-301
    rgbs, sig = sigma_rgb[..., :3], sigma_rgb[..., 3]
    alphas = 1 - torch.exp(-deltas * torch.relu(sig))                            # :344
    shifted = torch.cat([torch.ones_like(alphas[:, :1]), 1 - alphas + 1e-10], -1)
    T = torch.cumprod(shifted, -1)                                               # :350
    wts = alphas * T[:, :-1]
    rgb = torch.sum(wts.unsqueeze(-1) * rgbs, -2)
    depth = torch.sum(wts * z, -1)
    if white_bkgd:
        rgb = rgb + 1 - wts.sum(dim=-1).unsqueeze(-1)                            # :357-360
    return wts, rgb, depth


def composite(scene: Scene, w: MLPWeights, rays, z, white_bkgd, eval_batch_size=100000):
    NR, K = z.shape
    pts = (rays[:, None, :3] + z.unsqueeze(-1) * rays[:, None, 3:6]).reshape(-1, 3)   # :304
    dirs = rays[:, None, 3:6].expand(-1, K, -1).reshape(-1, 3)
    outs = [pixelnerf_forward(scene, w, p, d)
            for p, d in zip(torch.split(pts, eval_batch_size), torch.split(dirs, eval_batch_size))]   # :328-333
    field = torch.cat(outs, 0).reshape(NR, K, 4)
    return composite_from_field(field, rays, z, white_bkgd) + (field,)


# --------------------------------------------------------------------------------------
# a10 renderer.forward                (nerf_renderer.py:399-424)
# --------------------------------------------------------------------------------------
def render(scene: Scene, w: MLPWeights, rays, n_samples, n_cand, n_gaussian, white_bkgd,
           noise_coarse, noise_gauss, noise_fill):
    z0 = sample_depthguided(scene, rays, n_samples, n_cand, n_gaussian, noise_coarse, noise_gauss)
    z = fill_up_uniform_samples(z0, rays, noise_fill)
    wts, rgb, depth, field = composite(scene, w, rays, z, white_bkgd)
    return dict(rgb=rgb, depth=depth, weights=wts, z=z, z_unfilled=z0, field=field)


# --------------------------------------------------------------------------------------
# either side of the path (used to build inputs): rays and normals
# --------------------------------------------------------------------------------------
def gen_rays(extr, intr, W, H, z_near, z_far):
    """(4,4),(3,3) -> (H*W, 8) rays [o, d, near, far], row-major pixels, centres at +0.5
    (cam_geometry.py:5-48).  Ray generation is NOT on the hot path; like the reference it goes through
    torch.matmul, whose last bit depends on the host's BLAS kernels, so the golden fixtures store the rays
    themselves instead of regenerating them."""
    focal = intr[[0, 1], [0, 1]]
    c = intr[[0, 1], [-1, -1]]
    ys, xs = torch.meshgrid(torch.arange(.5, H, 1), torch.arange(.5, W, 1), indexing="ij")
    pc = (torch.stack((xs, ys), dim=-1) - c.view(1, 1, 2)) / focal.view(1, 1, 2)
    pc = torch.cat((pc, torch.ones_like(pc[..., :1])), dim=-1)
    dcam = pc / pc.pow(2).sum(dim=-1, keepdim=True).sqrt()
    Rc2w = extr[:3, :3].permute(1, 0)
    dw = (Rc2w @ dcam.view(-1, 3).permute(1, 0)).permute(1, 0)                   # cam_geometry.py:37-38
    o = (-1 * Rc2w @ extr[:3, -1:]).view(1, 3).expand(H * W, -1)                 # :41
    nf = torch.tensor([z_near, z_far], dtype=torch.float32).view(1, 2).expand(H * W, -1)
    return torch.cat((o, dw, nf), dim=-1)
