"""G19: golden vectors for configurations OUTSIDE the fused field kernels, from the IMPORTED reference (build container only).

    python oracle/make_golden_generic.py        # writes tests/golden/g19_generic.npz

SURVEY.md section 8(c) asked for "a reduced d_hidden / d_latent variant" of the MLP fixture; VERDICT r4 for ResnetFC with the reference's
own constructor defaults.  The reference's modules are run as they are (resnetfc.py:72-159, pixelnerf.py:55-145, nerf_renderer.py:399-424):
  A  ResnetFC(d_in=55, d_latent=512) with the constructor DEFAULTS (d_hidden 128, 5 blocks, combine_layer 1000: the views are never
     averaged -> (SB, NV, B, 4)), NV = 4
  B  ResnetFC(d_in=23, d_latent=40, d_out=5, d_hidden=128, n_blocks=3, combine_layer=2), NV = 3, SB = 2
  C  ResnetFC(d_in=20, d_latent=0, d_hidden=256, n_blocks=2, combine_layer=1, beta=1.5) (Softplus, no latent), NV = 2
  E  ResnetFC(d_in=7, d_latent=5, n_blocks=2, combine_layer=0) (views averaged before block 0; latent columns without a lin_z layer), NV = 3
  D  PixelNeRF with poscode num_freqs=4 / freq_factor=3.0, SpatialEncoder(num_layers=2) (latent width 128), the MLP
     n_blocks=3 / d_hidden=128 / combine_layer=2, NV = 3 source views: forward(xyz, viewdirs) and renderer.forward on 192 rays
     (K = 32, G = 12, injected noise)
Inputs are regenerated from seeds by the tests (sha256 guards); the reference's outputs are stored."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle.ref_import import import_reference, Conf                     # noqa: E402
from oracle.make_golden import inject_noise, sha                         # noqa: E402
from oracle import diner_oracle as O                                     # noqa: E402
from diner_amd.synthetic import make_scene, make_mlp_state_dict          # noqa: E402
from src.util.depth2normal import depth2normal                           # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")

MLP_CASES = {      # name: (constructor kwargs, NV, SB, B, seed)
    "A": (dict(d_in=55, d_latent=512), 4, 1, 64, 501),
    "B": (dict(d_in=23, d_latent=40, d_out=5, d_hidden=128, n_blocks=3, combine_layer=2), 3, 2, 50, 502),
    "C": (dict(d_in=20, d_latent=0, d_hidden=256, n_blocks=2, combine_layer=1, beta=1.5), 2, 1, 33, 503),
    "E": (dict(d_in=7, d_latent=5, n_blocks=2, combine_layer=0), 3, 1, 20, 505),      # views averaged BEFORE block 0: latent columns present, no lin_z layer
}
PIX = dict(W=48, H=40, nv=3, latent_ch=128, seed=7, num_freqs=4, freq_factor=3.0,
           mlp=dict(n_blocks=3, d_hidden=128, combine_layer=2, combine_type="average"), K=32, G=12, n_cand=1000, NR=192, B=300)


def mlp_state_dict(kw, seed):
    full = dict(d_in=0, d_out=4, n_blocks=5, d_latent=0, d_hidden=128, combine_layer=1000)
    full.update({k: v for k, v in kw.items() if k in full})
    sd = make_mlp_state_dict(seed=seed, d_in=max(full["d_in"], 1), d_latent=max(full["d_latent"], 1), d_hidden=full["d_hidden"],
                             d_out=full["d_out"], n_blocks=full["n_blocks"], combine_layer=full["combine_layer"])
    if full["d_latent"] == 0:
        sd = {k: v for k, v in sd.items() if not k.startswith("lin_z.")}
    return sd


def mlp_inputs(kw, nv, SB, B, seed):
    g = torch.Generator().manual_seed(seed + 1000)
    return torch.randn(SB, nv, B, kw.get("d_latent", 0) + kw["d_in"], generator=g)


def pix_scene(rays_fixture=None):
    """rays_fixture: the rays stored in the fixture (ray generation goes through torch.matmul, whose last bit depends on the host's BLAS
    kernels -- as for every other fixture the rays themselves are stored)."""
    sc = make_scene(PIX["W"], PIX["H"], nv=PIX["nv"], seed=PIX["seed"], latent_ch=PIX["latent_ch"])
    sc["normals"] = depth2normal(sc["depths"], sc["src_intrinsics"])
    per = 2 * PIX["num_freqs"] + 1
    msd = make_mlp_state_dict(seed=77, d_in=4 * per + 3, d_latent=PIX["latent_ch"], d_hidden=PIX["mlp"]["d_hidden"], d_out=4,
                              n_blocks=PIX["mlp"]["n_blocks"], combine_layer=PIX["mlp"]["combine_layer"])
    rays = O.gen_rays(sc["target_extrinsics"], sc["target_intrinsics"], PIX["W"], PIX["H"], sc["znear"], sc["zfar"])
    g = torch.Generator().manual_seed(78)
    sel = torch.randperm(rays.shape[0], generator=g)[:PIX["NR"]].sort().values
    rays = rays[sel].contiguous() if rays_fixture is None else torch.as_tensor(rays_fixture)
    noise = (torch.rand(PIX["NR"], PIX["n_cand"], generator=g), torch.randn(PIX["NR"], PIX["G"], generator=g),
             torch.rand(PIX["NR"], PIX["K"], generator=g))
    z = sc["znear"] + (sc["zfar"] - sc["znear"]) * torch.rand(PIX["B"], generator=g)
    pick = torch.randint(0, PIX["NR"], (PIX["B"],), generator=g)
    xyz = rays[pick, :3] + z[:, None] * rays[pick, 3:6]
    dirs = rays[pick, 3:6].contiguous()
    return sc, msd, rays, noise, xyz, dirs


def main():
    torch.manual_seed(0)
    ns = import_reference()
    out = {}
    with torch.no_grad():
        for name, (kw, nv, SB, B, seed) in MLP_CASES.items():
            m = ns.resnetfc.ResnetFC(**kw)
            m.load_state_dict(mlp_state_dict(kw, seed), strict=True)
            zx = mlp_inputs(kw, nv, SB, B, seed)
            y = m(zx, combine_dim=1)
            out[f"mlp{name}_out"] = y.numpy()
            out[f"mlp{name}_in_sha"] = np.array(sha(zx))
            print(f"G19 ResnetFC case {name}: zx {tuple(zx.shape)} -> {tuple(y.shape)}, |out| max {float(y.abs().max()):.3f}")
        # ---- D: PixelNeRF / renderer in a non-shipped configuration
        import sys as _s
        saved = {k: v for k, v in _s.modules.items() if k == "src" or k.startswith("src.")}
        for k in saved:
            del _s.modules[k]
        _s.modules.update(ns._modules)
        try:
            nerf = ns.pixelnerf.PixelNeRF(
                poscode_conf=Conf(kwargs=dict(num_freqs=PIX["num_freqs"], freq_factor=PIX["freq_factor"], include_input=True)),
                encoder_conf=Conf(module="src.models.image_encoder.SpatialEncoder",
                                  kwargs=dict(image_padding=64, padding_pe=4, pretrained=False, num_layers=2)),
                mlp_fine_conf=Conf(module="src.models.resnetfc.ResnetFC", kwargs=PIX["mlp"]))
        finally:
            for k in [k for k in _s.modules if k == "src" or k.startswith("src.")]:
                del _s.modules[k]
            _s.modules.update(saved)
        nerf = nerf.eval()
        sc, msd, rays, noise, xyz, dirs = pix_scene()
        assert nerf.d_latent == PIX["latent_ch"] and nerf.d_in == 4 * (2 * PIX["num_freqs"] + 1) + 3
        nerf.mlp_fine.load_state_dict(msd, strict=True)
        enc = nerf.encoder
        enc.depths, enc.depths_std, enc.normals = sc["depths"][None], sc["depths_std"][None], sc["normals"][None]
        enc.latent = sc["latent"][None]
        enc.nviews, enc.nobjects = PIX["nv"], 1
        nerf.poses = sc["src_extrinsics"][None]
        nerf.c = sc["src_intrinsics"][None, :, :2, -1]
        nerf.focal = sc["src_intrinsics"][None][:, :, [0, 1], [0, 1]]
        nerf.image_shape = sc["image_shape"].clone()
        f = nerf(xyz[None], viewdirs=dirs[None])
        out["pix_field"] = f[0].numpy()
        ren = ns.nerf_renderer.NeRFRendererDGS(n_samples=PIX["K"], n_depth_candidates=PIX["n_cand"], n_gaussian=PIX["G"], white_bkgd=False)
        with inject_noise(*noise):
            res = ren.forward(nerf, rays[None], want_weights=True)
            z = ren.fill_up_uniform_samples(ren.sample_depthguided(rays[None], nerf, PIX["K"], PIX["n_cand"], n_gaussian=PIX["G"]), rays[None])
        out["pix_rgb"], out["pix_depth"], out["pix_z"] = res.fine.rgb[0].numpy(), res.fine.depth[0].numpy(), z[0].numpy()
        out["pix_rays"] = rays.numpy()
        out["pix_in_sha"] = np.array(sha(xyz, dirs, *noise))
        print(f"G19 PixelNeRF (NV 3, latent 128, num_freqs 4, d_hidden 128, 3 blocks, combine 2): field {tuple(f.shape)}, rgb {tuple(res.fine.rgb.shape)}")
    np.savez_compressed(os.path.join(OUT, "g19_generic.npz"), **out)
    print("wrote", os.path.join(OUT, "g19_generic.npz"), os.path.getsize(os.path.join(OUT, "g19_generic.npz")), "bytes")


if __name__ == "__main__":
    main()
