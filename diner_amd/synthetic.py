"""Synthetic 4-view scenes for parity tests and bench.py.

There is no DTU / Facescape data and no checkpoint in the build or bench environment,
so the workload is an analytic scene: a sphere (r = 0.25) in front of a finite back plane,
seen by NV source cameras on a +-20 degree arc of radius 1 and one target camera at the
arc centre.  Depth maps are rendered analytically (background depth 0), the depth
standard deviation follows the DTU confidence law of the reference (dtu.py:68-70:
std = 0.0328 - 0.0257 * conf), the latent feature map is seeded N(0,1) (the ResNet34
trunk is not part of the hot path) and the MLP keeps the reference initialisation except
for fc_1, which the reference zero-initialises (resnetfc.py:47) and which would hide
MLP bugs if left at zero.

Everything here is plain torch on the CPU (deterministic given the seed); callers move
the tensors to the GPU.
"""
import math

import torch


def look_at_extrinsics(cam_pos, target=(0.0, 0.0, 0.0), up=(0.0, -1.0, 0.0)):
    """world->camera (4,4), OpenCV convention (x right, y down, z forward).  Pure Python float math so that
    the matrix is bit-identical on every host (torch reductions are not)."""
    def sub(a, b): return [a[i] - b[i] for i in range(3)]
    def cross(a, b): return [a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]]
    def unit(a):
        n = math.sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2])
        return [a[0] / n, a[1] / n, a[2] / n]
    p = [float(v) for v in cam_pos]
    z = unit(sub([float(v) for v in target], p))
    x = unit(cross(z, [float(v) for v in up]))
    y = cross(z, x)
    R = [x, y, z]                       # rows = camera axes in world coordinates
    E = [[R[i][0], R[i][1], R[i][2], -(R[i][0] * p[0] + R[i][1] * p[1] + R[i][2] * p[2])] for i in range(3)]
    E.append([0.0, 0.0, 0.0, 1.0])
    return torch.tensor(E, dtype=torch.float64).float()


def _analytic_depth(E, Kmat, W, H, radius=0.25, plane_z=0.35, plane_half=0.6):
    """z-depth (camera frame) of the first hit of the sphere / finite back plane; 0 = background.
    Element-wise float64 ops only (IEEE, host independent)."""
    E = E.double()
    Kd = Kmat.double()
    ys, xs = torch.meshgrid(torch.arange(0.5, H, 1.0, dtype=torch.float64),
                            torch.arange(0.5, W, 1.0, dtype=torch.float64), indexing="ij")
    dcx, dcy = (xs - Kd[0, 2]) / Kd[0, 0], (ys - Kd[1, 2]) / Kd[1, 1]
    R, t = E[:3, :3], E[:3, 3]
    o = [-(R[0, i] * t[0] + R[1, i] * t[1] + R[2, i] * t[2]) for i in range(3)]          # -R^T t
    dw = [R[0, i] * dcx + R[1, i] * dcy + R[2, i] for i in range(3)]                     # R^T [dcx, dcy, 1]
    # sphere |o + s d|^2 = r^2   (s is the camera-frame z-depth because the camera-frame direction has z == 1)
    a = dw[0] * dw[0] + dw[1] * dw[1] + dw[2] * dw[2]
    b = 2 * (dw[0] * o[0] + dw[1] * o[1] + dw[2] * o[2])
    c = (o[0] * o[0] + o[1] * o[1] + o[2] * o[2]) - radius ** 2
    disc = b * b - 4 * a * c
    inf = torch.full_like(a, float("inf"))
    s_sph = torch.where(disc > 0, (-b - torch.sqrt(disc.clamp(min=0))) / (2 * a), inf)
    s_sph = torch.where(s_sph > 0, s_sph, inf)
    # plane z_world = plane_z, finite extent
    s_pl = (plane_z - o[2]) / dw[2]
    hx, hy = o[0] + s_pl * dw[0], o[1] + s_pl * dw[1]
    ok = (s_pl > 0) & (hx.abs() <= plane_half) & (hy.abs() <= plane_half)
    s_pl = torch.where(ok, s_pl, inf)
    s = torch.minimum(s_sph, s_pl)
    return torch.where(torch.isfinite(s), s, torch.zeros_like(s)).float()


def make_scene(W=64, H=64, nv=4, seed=0, latent_ch=512, image_padding=64, znear=0.5, zfar=1.5,
               bg_std_zero=False, latent=True, scale=1.0, std_law="dtu"):
    """Returns a dict of CPU tensors describing one object:
        src_extrinsics (NV,4,4), src_intrinsics (NV,3,3), depths / depths_std (NV,1,H,W),
        latent (NV,C,Hf,Wf) [if latent], target_extrinsics (4,4), target_intrinsics (3,3),
        image_shape (2,) = [W,H], znear, zfar, feature_padding.
    Normals are NOT included: they are derived from the depth maps by depth2normal at encode time.
    `scale` multiplies the whole geometry (camera distance, sphere, plane): scale=1.75 with znear/zfar = 1.0/2.5 is
    the Facescape depth range (facescape.py:19-20); std_law="facescape" uses that dataset's confidence law
    (facescape.py:50-52: std = 1.649e-2 - 1.582e-2 * conf).
    """
    g = torch.Generator().manual_seed(seed)
    Kmat = torch.tensor([[1.2 * W, 0.0, W / 2.0], [0.0, 1.2 * W, H / 2.0], [0.0, 0.0, 1.0]])
    angs = torch.linspace(-20.0, 20.0, nv) if nv > 1 else torch.zeros(1)
    extr = []
    for i, a in enumerate(angs.tolist()):
        th = math.radians(a)
        elev = 0.06 * ((-1) ** i)
        extr.append(look_at_extrinsics((scale * math.sin(th), scale * elev, -scale * math.cos(th))))
    extr = torch.stack(extr)
    intr = Kmat.unsqueeze(0).repeat(nv, 1, 1).clone()
    depths = torch.stack([_analytic_depth(extr[i], intr[i], W, H, radius=0.25 * scale, plane_z=0.35 * scale,
                                          plane_half=0.6 * scale) for i in range(nv)]).unsqueeze(1)
    conf = torch.rand(nv, 1, H, W, generator=g) * 0.7 + 0.3
    if std_law == "dtu":
        std = 0.0328 - 0.0257 * conf                   # dtu.py:68-70
    elif std_law == "facescape":
        std = 1.649e-2 - 1.582e-2 * conf               # facescape.py:50-52
    else:
        raise ValueError(f"unknown std_law {std_law!r}")
    if bg_std_zero:
        std = torch.where(depths == 0, torch.zeros_like(std), std)   # multiface.py:310 variant
    out = dict(src_extrinsics=extr, src_intrinsics=intr, depths=depths, depths_std=std,
               target_extrinsics=look_at_extrinsics((0.03 * scale, -0.02 * scale, -1.0 * scale)),
               target_intrinsics=Kmat.clone(),
               image_shape=torch.tensor([float(W), float(H)]), znear=znear, zfar=zfar,
               feature_padding=image_padding / 2.0, W=W, H=H)
    if latent:
        Hf, Wf = (H + 2 * image_padding) // 2, (W + 2 * image_padding) // 2
        out["latent"] = torch.randn(nv, latent_ch, Hf, Wf, generator=g)
    return out


def randomize_mlp_(mlp, seed=1234, fc1_std=0.03, bias_std=0.05):
    """Keep the reference init (kaiming fan_in weights) but give fc_1 and the biases non-zero values."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for blk in mlp.blocks:
            blk.fc_1.weight.copy_(torch.randn(blk.fc_1.weight.shape, generator=g) * fc1_std)
            blk.fc_0.bias.copy_(torch.randn(blk.fc_0.bias.shape, generator=g) * bias_std)
            blk.fc_1.bias.copy_(torch.randn(blk.fc_1.bias.shape, generator=g) * bias_std)
        for lz in mlp.lin_z:
            lz.bias.copy_(torch.randn(lz.bias.shape, generator=g) * bias_std)
        mlp.lin_in.bias.copy_(torch.randn(mlp.lin_in.bias.shape, generator=g) * bias_std)
        mlp.lin_out.bias.copy_(torch.randn(mlp.lin_out.bias.shape, generator=g) * bias_std)
    return mlp


def make_mlp_state_dict(seed=1234, d_in=55, d_latent=512, d_hidden=512, d_out=4, n_blocks=5,
                        combine_layer=3, fc1_std=0.03, bias_std=0.05):
    """ResnetFC state_dict (reference key names, resnetfc.py:72-127) drawn from an explicit CPU
    generator: kaiming-normal(fan_in) weights like the reference init, but non-zero fc_1 and biases."""
    g = torch.Generator().manual_seed(seed)

    def kaiming(o, i):
        return torch.randn(o, i, generator=g) * math.sqrt(2.0 / i)

    def bias(o):
        return torch.randn(o, generator=g) * bias_std

    sd = {"lin_in.weight": kaiming(d_hidden, d_in), "lin_in.bias": bias(d_hidden),
          "lin_out.weight": kaiming(d_out, d_hidden), "lin_out.bias": bias(d_out)}
    for b in range(n_blocks):
        sd[f"blocks.{b}.fc_0.weight"] = kaiming(d_hidden, d_hidden)
        sd[f"blocks.{b}.fc_0.bias"] = bias(d_hidden)
        sd[f"blocks.{b}.fc_1.weight"] = torch.randn(d_hidden, d_hidden, generator=g) * fc1_std
        sd[f"blocks.{b}.fc_1.bias"] = bias(d_hidden)
    for b in range(min(combine_layer, n_blocks)):
        sd[f"lin_z.{b}.weight"] = kaiming(d_hidden, d_latent)
        sd[f"lin_z.{b}.bias"] = bias(d_hidden)
    return sd


class _Conf:
    """Stand-in for the omegaconf nodes the reference hands to its constructors (`.module`, `.kwargs`)."""

    def __init__(self, module=None, kwargs=None):
        self.module, self.kwargs = module, (kwargs or {})


def build_modules(sc, msd, device, normals=None):
    """The drop-in modules exactly as the reference builds them (import_obj of dotted paths, diner.py:47-48 /
    pixelnerf.py:17-24), with the synthetic scene `sc` (or a LIST of scenes of one size = the SB objects of a training batch,
    configs/train_dtu.yaml:16) injected into the encoder (feature maps and cameras set directly, no
    ResNet34 weights involved) and the MLP state dict `msd` loaded strictly.
    -> (PixelNeRF on `device`, NeRFRendererDGS class)."""
    import torch as _t
    from src.util.import_helper import import_obj
    nerf = import_obj("src.models.pixelnerf.PixelNeRF")(
        poscode_conf=_Conf(kwargs=dict(num_freqs=6, freq_factor=6.28, include_input=True)),
        encoder_conf=_Conf("src.models.image_encoder.SpatialEncoder", dict(image_padding=64, padding_pe=4, pretrained=False)),
        mlp_fine_conf=_Conf("src.models.resnetfc.ResnetFC", dict(n_blocks=5, d_hidden=512, combine_layer=3, combine_type="average")))
    nerf.mlp_fine.load_state_dict(msd, strict=True)
    nerf = nerf.to(device).eval()
    enc = nerf.encoder
    scs = list(sc) if isinstance(sc, (list, tuple)) else [sc]
    if normals is None:
        from src.util.depth2normal import depth2normal
        normals = [depth2normal(s["depths"].to(device), s["src_intrinsics"].to(device)) for s in scs]
    elif not isinstance(normals, (list, tuple)):
        normals = [normals]
    st = lambda key: _t.stack([s[key] for s in scs]).to(device)
    enc.depths, enc.depths_std = st("depths"), st("depths_std")
    enc.normals, enc.latent = _t.stack([n.to(device) for n in normals]), st("latent")
    enc.nviews, enc.nobjects = int(scs[0]["depths"].shape[0]), len(scs)
    Kin = st("src_intrinsics")
    nerf.poses = st("src_extrinsics")
    nerf.c = Kin[:, :, :2, -1].contiguous()
    nerf.focal = Kin[:, :, [0, 1], [0, 1]].contiguous()
    nerf.image_shape = scs[0]["image_shape"].clone().to(device)
    return nerf, import_obj("src.models.nerf_renderer.NeRFRendererDGS")


# ---- "realistic magnitudes" (round 6, fixture G20): what a trained checkpoint's tensors look like, as far as magnitudes go --------------
# ResNet34 features are non-negative and heavy-tailed and trained weights are not Kaiming-distributed; the f16x3 split of the field kernels
# (|w| < 1024, activations below the fp16 range, DESIGN.md "Arithmetic modes") was exercised by N(0,1) latents and init-distributed weights
# only.  Both recipes use exactly rounded element-wise operations on seeded normal / uniform draws (powers of two, multiplications,
# additions), so that they are bit-identical on every host; the fixture pins a sha256 of what they produce.
def realistic_latent(nv, C, Hf, Wf, seed, hot_gain=8.0, n_hot=6):
    """(nv, C, Hf, Wf): relu(N(0,1)) x a per-channel power-of-two scale 2^round(1.2 N(0,1)) (clamped to [1/8, 16]) x a heavy element tail
    (1 + |N(0,1)|^3 / 8), and n_hot channels x hot_gain (a few dominant feature channels): mean 0.74, 99.9 % below 30, maximum ~470 at the defaults."""
    g = torch.Generator().manual_seed(seed)
    x = torch.relu(torch.randn(nv, C, Hf, Wf, generator=g))
    n = torch.round(1.2 * torch.randn(1, C, 1, 1, generator=g)).clamp(-3, 4)
    x = x * torch.ldexp(torch.ones_like(n), n.to(torch.int32))
    t = torch.randn(nv, C, Hf, Wf, generator=g).abs()
    x = x * (1.0 + 0.125 * (t * t * t))
    hot = torch.randperm(C, generator=g)[:n_hot]
    x[:, hot] = x[:, hot] * hot_gain
    return x.contiguous()


def realistic_mlp_state_dict(seed=4321, n_planted=3, planted_lo=10.0, planted_hi=50.0, bias_std=1.0):
    """make_mlp_state_dict's tensors with row-wise power-of-two scales 2^round(0.8 N(0,1)) (clamped to [1/4, 8]), n_planted entries per weight
    matrix of magnitude planted_lo .. planted_hi (random sign) and biases of O(1)."""
    sd = make_mlp_state_dict(seed=seed)
    g = torch.Generator().manual_seed(seed + 1)
    out = {}
    for k in sorted(sd):
        v = sd[k].clone()
        if k.endswith(".weight"):
            o, i = v.shape
            n = torch.round(0.8 * torch.randn(o, 1, generator=g)).clamp(-2, 3)
            if not k.startswith("lin_out"):
                v = v * torch.ldexp(torch.ones_like(n), n.to(torch.int32))
                idx = torch.randint(0, o * i, (n_planted,), generator=g)
                mag = planted_lo + (planted_hi - planted_lo) * torch.rand(n_planted, generator=g)
                sgn = torch.where(torch.rand(n_planted, generator=g) < 0.5, -torch.ones(n_planted), torch.ones(n_planted))
                v.view(-1)[idx] = mag * sgn
        else:
            v = torch.randn(v.shape, generator=g) * bias_std
        out[k] = v.contiguous()
    return out


def as_encoded(latent):
    """A (SB, NV, C, Hf, Wf) latent in the memory format `PixelNeRF.encode` of this repo emits on a HIP device: NCHW shape, channels-last
    strides (src/models/image_encoder.py concatenates the pyramid in that format).  Synthetic leaf latents of the training timers use it so
    that the timed step sees the latent the way a real encoder hands it over."""
    import torch as _t
    return latent.flatten(0, 1).contiguous(memory_format=_t.channels_last).view(latent.shape)
