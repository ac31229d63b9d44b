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
    # training through a non-shipped configuration is refused, not silently wrong
    nerf.train()
    for p in nerf.mlp_fine.parameters():
        p.requires_grad_(True)
    with pytest.raises(NotImplementedError):
        nerf(xyz[None].cuda(), viewdirs=dirs[None].cuda())
