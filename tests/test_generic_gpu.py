"""GPU: configurations OUTSIDE the fused field kernels run on the generic slow path (csrc/generic.hip, ABI v5) and match the reference's
own modules (G19, oracle/make_golden_generic.py): "same constructor kwargs" is "same behaviour" (resnetfc.py:72-159, pixelnerf.py:13-145).
Until round 5 these constructed and then failed at the first call (DINER_E_UNSUPPORTED / NotImplementedError)."""
import numpy as np
import pytest
import torch

from oracle.make_golden_generic import MLP_CASES, PIX, mlp_state_dict, mlp_inputs, pix_scene
from tests.helpers import load, max_norm_rel, sha

pytestmark = pytest.mark.gpu
TOL = 1e-4


def T(a):
    return torch.from_numpy(np.asarray(a))


@pytest.mark.parametrize("name", sorted(MLP_CASES))
def test_resnetfc_any_configuration(name):
    """A: the constructor DEFAULTS of the reference (d_hidden 128, combine_layer 1000: no view fusion inside the network);
    B: d_hidden 128 / 3 blocks / combine 2 / NV 3 / d_out 5 / SB 2; C: Softplus (beta 1.5), no latent, NV 2."""
    from diner_amd import ops
    from src.models.resnetfc import ResnetFC
    g = load("g19_generic.npz")
    kw, nv, SB, B, seed = MLP_CASES[name]
    m = ResnetFC(**kw)
    m.load_state_dict(mlp_state_dict(kw, seed), strict=True)
    m = m.cuda().eval()
    zx = mlp_inputs(kw, nv, SB, B, seed)
    assert sha(zx) == str(g[f"mlp{name}_in_sha"]), "seeded inputs not reproducible on this host"
    with torch.no_grad():
        y = m(zx.cuda(), combine_dim=1)
    want = T(g[f"mlp{name}_out"])
    assert isinstance(m.hip_mlp(nv=nv), ops.GenericMlp)
    assert tuple(y.shape) == tuple(want.shape)
    e = max_norm_rel(y.cpu(), want)
    print(f"ResnetFC case {name} {kw}: {tuple(y.shape)}, max-norm-rel {e:.2e}")
    assert e < 2e-5
    with torch.no_grad():                      # the (NV, B, C) / combine_dim=0 form of the same call
        y0 = m(zx[0].cuda(), combine_dim=0)
    assert torch.equal(y0, y[0])


def test_shipped_shape_with_three_views_takes_the_generic_path():
    """The fused kernels are built for NV = 4; the shipped MLP on three views must still run (and agree with the oracle's restatement)."""
    from diner_amd import ops
    from diner_amd.synthetic import make_mlp_state_dict
    from oracle import diner_oracle as O
    from src.models.resnetfc import ResnetFC
    msd = make_mlp_state_dict()
    m = ResnetFC(d_in=55, d_latent=512, n_blocks=5, d_hidden=512, combine_layer=3)
    m.load_state_dict(msd, strict=True)
    m = m.cuda().eval()
    zx = torch.randn(3, 40, 567, generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        y = m(zx.cuda(), combine_dim=0)
    assert isinstance(m.hip_mlp(nv=3), ops.GenericMlp) and isinstance(m.hip_mlp(nv=4), ops.HipMlp)
    want = O.mlp_forward(O.MLPWeights.from_state_dict(msd), zx)
    assert max_norm_rel(y.cpu(), want) < 2e-5


def _pix_modules(sc, msd):
    from diner_amd.synthetic import _Conf
    from src.util.import_helper import import_obj
    nerf = import_obj("src.models.pixelnerf.PixelNeRF")(
        poscode_conf=_Conf(kwargs=dict(num_freqs=PIX["num_freqs"], freq_factor=PIX["freq_factor"], include_input=True)),
        encoder_conf=_Conf("src.models.image_encoder.SpatialEncoder", dict(image_padding=64, padding_pe=4, pretrained=False, num_layers=2)),
        mlp_fine_conf=_Conf("src.models.resnetfc.ResnetFC", PIX["mlp"]))
    nerf.mlp_fine.load_state_dict(msd, strict=True)
    nerf = nerf.cuda().eval()
    enc = nerf.encoder
    enc.depths, enc.depths_std = sc["depths"][None].cuda(), sc["depths_std"][None].cuda()
    enc.normals, enc.latent = sc["normals"][None].cuda(), sc["latent"][None].cuda()
    enc.nviews, enc.nobjects = PIX["nv"], 1
    Kin = sc["src_intrinsics"]
    nerf.poses = sc["src_extrinsics"][None].cuda()
    nerf.c = Kin[None, :, :2, -1].contiguous().cuda()
    nerf.focal = Kin[None][:, :, [0, 1], [0, 1]].contiguous().cuda()
    nerf.image_shape = sc["image_shape"].clone().cuda()
    return nerf, import_obj("src.models.nerf_renderer.NeRFRendererDGS")


def test_pixelnerf_and_renderer_in_a_non_shipped_configuration():
    """D: poscode num_freqs 4 / freq_factor 3.0 (d_in 39), latent width 128 (SpatialEncoder num_layers=2), MLP 3 blocks x 128 hidden,
    combine 2, THREE source views -- PixelNeRF.forward, renderer.composite on the reference's samples (every ray) and renderer.forward with
    injected noise, against the reference's outputs."""
    from diner_amd import noise
    g = load("g19_generic.npz")
    sc, msd, rays, nz, xyz, dirs = pix_scene(g["pix_rays"])
    assert sha(xyz, dirs, *nz) == str(g["pix_in_sha"]), "seeded inputs not reproducible on this host"
    nerf, R = _pix_modules(sc, msd)
    assert nerf.is_generic() and nerf.d_in == 39 and nerf.d_latent == 128
    with torch.no_grad():
        f = nerf(xyz[None].cuda(), viewdirs=dirs[None].cuda())
    e_f = max_norm_rel(f[0].cpu(), g["pix_field"])
    ren = R(n_samples=PIX["K"], n_depth_candidates=PIX["n_cand"], n_gaussian=PIX["G"], white_bkgd=False)
    rc = T(g["pix_rays"]).cuda()[None]
    ref_rgb, ref_d, ref_z = T(g["pix_rgb"]), T(g["pix_depth"]), T(g["pix_z"])
    with torch.no_grad():
        wts, rgb, depth = ren.composite(nerf, rc, ref_z.cuda()[None])
    e_rgb = ((rgb[0].cpu() - ref_rgb).abs().max(-1).values / ref_rgb.abs().max()).max().item()
    e_d = ((depth[0].cpu() - ref_d).abs() / ref_d.abs().max()).max().item()
    with torch.no_grad(), noise.inject(*(t.cuda()[None] for t in nz)):
        out = ren.forward(nerf, rc)
        z = ren.fill_up_uniform_samples(ren.sample_depthguided(rc, nerf, PIX["K"], PIX["n_cand"], n_gaussian=PIX["G"]), rc)
    same = torch.isclose(z[0].cpu(), ref_z, rtol=3e-6, atol=1e-7).all(-1)
    s_rgb = ((out.fine.rgb[0].cpu() - ref_rgb).abs().max(-1).values / ref_rgb.abs().max())[same].max().item()
    s_d = ((out.fine.depth[0].cpu() - ref_d).abs() / ref_d.abs().max())[same].max().item()
    print(f"generic PixelNeRF: field {e_f:.2e}; composite on the reference's samples rgb {e_rgb:.2e} depth {e_d:.2e} (all {PIX['NR']} rays); "
          f"renderer.forward: {int(same.sum())}/{PIX['NR']} rays with the reference's sample set, rgb {s_rgb:.2e} depth {s_d:.2e}")
    assert e_f < 2e-5 and e_rgb < TOL and e_d < TOL
    assert int(same.sum()) >= PIX["NR"] - 4 and s_rgb < TOL and s_d < TOL
    # round 6: training through a non-shipped configuration -- gradients of PixelNeRF.forward with respect to the MLP parameters and the
    # encoder's latent against torch autograd through the CPU oracle (the restatement covers this configuration bit-exactly, test_oracle_golden)
    from oracle import diner_oracle as O
    from tests.tests_train_util import oracle_key
    nerf.train()
    for p in nerf.mlp_fine.parameters():
        p.requires_grad_(True)
    nerf.encoder.latent = nerf.encoder.latent.detach().requires_grad_(True)
    Gm = torch.randn(xyz.shape[0], 4, generator=torch.Generator().manual_seed(3))
    out = nerf(xyz[None].cuda(), viewdirs=dirs[None].cuda())
    assert out.requires_grad and max_norm_rel(out[0].detach().cpu(), g["pix_field"]) < 2e-5
    (out[0] * Gm.cuda()).sum().backward()
    Kin = sc["src_intrinsics"]
    lat = sc["latent"].clone().requires_grad_(True)
    scene = O.Scene(latent=lat, depths=sc["depths"], depths_std=sc["depths_std"], normals=sc["normals"], poses=sc["src_extrinsics"],
                    focal=Kin[:, [0, 1], [0, 1]], c=Kin[:, :2, -1], image_shape=sc["image_shape"], feature_padding=sc["feature_padding"])
    w = O.MLPWeights.from_state_dict(msd, combine_layer=PIX["mlp"]["combine_layer"], d_latent=PIX["latent_ch"])
    leaves = {}
    for k, v in vars(w).items():
        for i, t in enumerate(v if isinstance(v, (list, tuple)) else [v]):
            if torch.is_tensor(t) and t.is_floating_point():
                leaves[(k, i if isinstance(v, (list, tuple)) else None)] = t.requires_grad_(True)
    raw = O.mlp_forward(w, O.mlp_input(scene, xyz, dirs, PIX["num_freqs"], PIX["freq_factor"]))
    f = torch.cat([torch.sigmoid(raw[..., :3]), torch.relu(raw[..., 3:4])], dim=-1)
    (f * Gm).sum().backward()
    worst = max(((n, max_norm_rel(p.grad.cpu(), leaves[oracle_key(n)].grad)) for n, p in nerf.mlp_fine.named_parameters()), key=lambda t: t[1])
    e_lat = max_norm_rel(nerf.encoder.latent.grad[0].cpu(), lat.grad)
    print(f"generic PixelNeRF in grad mode: worst parameter gradient {worst[0]} {worst[1]:.2e}, d latent {e_lat:.2e} (max-norm-rel against the oracle's autograd)")
    assert worst[1] < TOL and e_lat < TOL
    # ... and through the renderer (sampler + field + compositor nodes): finite gradients on every parameter and the latent
    for p in nerf.mlp_fine.parameters():
        p.grad = None
    nerf.encoder.latent.grad = None
    with noise.inject(*(t.cuda()[None] for t in nz)):
        o2 = ren.forward(nerf, rc)
    o2.fine.rgb.sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in nerf.mlp_fine.parameters())
    assert torch.isfinite(nerf.encoder.latent.grad).all() and float(nerf.encoder.latent.grad.abs().max()) > 0


def _ref_resnetfc64(zx, sd, kw):
    """float64 torch restatement of ResnetFC.forward (resnetfc.py:129-159) for the gradient check: ReLU or Softplus(beta), mean over the views
    at combine_layer, lin_z on the blocks in front of it."""
    import torch.nn.functional as F
    full = dict(d_in=0, d_out=4, n_blocks=5, d_latent=0, d_hidden=128, combine_layer=1000, beta=0.0)
    full.update(kw)
    act = (lambda t: F.softplus(t, beta=full["beta"])) if full["beta"] > 0 else torch.relu
    z, x_in = zx[..., :full["d_latent"]], zx[..., full["d_latent"]:]
    x = F.linear(x_in, sd["lin_in.weight"], sd["lin_in.bias"]) if full["d_in"] > 0 else torch.zeros(*zx.shape[:-1], full["d_hidden"], dtype=zx.dtype)
    for b in range(full["n_blocks"]):
        if b == full["combine_layer"]:
            x = x.mean(0)
        if full["d_latent"] > 0 and b < full["combine_layer"]:
            x = x + F.linear(z, sd[f"lin_z.{b}.weight"], sd[f"lin_z.{b}.bias"])
        net = F.linear(act(x), sd[f"blocks.{b}.fc_0.weight"], sd[f"blocks.{b}.fc_0.bias"])
        x = x + F.linear(act(net), sd[f"blocks.{b}.fc_1.weight"], sd[f"blocks.{b}.fc_1.bias"])
    return F.linear(act(x), sd["lin_out.weight"], sd["lin_out.bias"])


@pytest.mark.parametrize("name", sorted(MLP_CASES))
def test_resnetfc_any_configuration_trains(name):
    """Round 6: ResnetFC.forward in grad mode for any configuration -- the constructor DEFAULTS (d_hidden 128, no view fusion),
    the reduced variant with a combine layer and three views, Softplus -- gradients of every parameter and of the input matrix against float64
    torch autograd of a restatement of resnetfc.py:129-159 (diner_mlp_generic_train_forward_f32 / _backward_f32: exact-fp32 GEMMs)."""
    from src.models.resnetfc import ResnetFC
    kw, nv, SB, B, seed = MLP_CASES[name]
    sd = mlp_state_dict(kw, seed)
    m = ResnetFC(**kw)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().train()
    zx = mlp_inputs(kw, nv, SB, B, seed)
    zg = zx.cuda().requires_grad_(True)
    y = m(zg, combine_dim=1)
    with torch.no_grad():
        m.eval()
        y_inf = m(zx.cuda(), combine_dim=1)
        m.train()
    assert max_norm_rel(y.detach().cpu(), y_inf.cpu()) < 1e-6            # the training forward is the inference forward
    Gm = torch.randn(y.shape, generator=torch.Generator().manual_seed(seed + 7))
    (y * Gm.cuda()).sum().backward()
    sd64 = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    z64 = zx.double().requires_grad_(True)
    y64 = torch.stack([_ref_resnetfc64(z64[i], sd64, kw) for i in range(SB)])
    assert max_norm_rel(y.detach().cpu(), y64.detach().float()) < 2e-5
    (y64 * Gm.double()).sum().backward()
    worst = max(((k, max_norm_rel(p.grad.cpu(), sd64[k].grad.float())) for k, p in m.named_parameters()), key=lambda t: t[1])
    e_z = max_norm_rel(zg.grad.cpu(), z64.grad.float())
    print(f"ResnetFC case {name} in grad mode: worst parameter gradient {worst[0]} {worst[1]:.2e}, d zx {e_z:.2e} (max-norm-rel against float64 autograd)")
    assert worst[1] < TOL and e_z < TOL
