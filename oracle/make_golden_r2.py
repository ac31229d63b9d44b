"""Round-2 golden vectors from the IMPORTED reference (build container only; test infrastructure).

    python oracle/make_golden_r2.py [g9] [g10] [g11] [g16]      # default: all four

G9   renderer.forward at the metric's sample count: K=128 / G=48 / 1000 candidates (create_prediction_folder.py:44-47
     with --nsamples 128), 64x64 lattice of rays of the 400x300 bench scene (BASELINE configs[1]), white_bkgd=False.
G10  renderer.forward in the Facescape evaluation configuration (BASELINE configs[4]): K=192 / G=72, white_bkgd=True,
     znear/zfar = 1.0/2.5 (facescape.py:19-20), facescape confidence->std law, 64x64 lattice of a 256x256 target
     (the processed Facescape image size, SURVEY.md "Key dimensions").
G16  (round 3) K=192 / G=72 like G10, but on the 400x300 bench scene with the DTU confidence->std law and range (48x48 lattice,
     black background): the wide DTU sigmas leave few candidates in the erf-saturation zone, which shows that the large share of
     implementation-defined picks in G10 belongs to the narrow Facescape sigmas, not to the sample count.
G11  the helpers either side of the path, run through the reference's own functions: torch_cmap (torch_helpers.py:42-75),
     depth2normal on maps with holes (depth2normal.py:7-87), gen_rays (cam_geometry.py:5-48).

As in make_golden.py: the reference runs with injected noise, the oracle restatement runs on the same inputs (every
8th ray) and must agree to fp32 round-off; the reference's outputs are what is stored.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import diner_oracle as O                                     # noqa: E402
from oracle.ref_import import import_reference, build_reference_nerf     # noqa: E402
from oracle.make_golden import inject_noise, report, sha, OUT            # noqa: E402
from diner_amd.synthetic import make_scene, make_mlp_state_dict          # noqa: E402


def lattice(W, H, n=64):
    rows = torch.linspace(0, H - 1, n).round().long()
    cols = torch.linspace(0, W - 1, n).round().long()
    return (rows[:, None] * W + cols[None, :]).reshape(-1)


def setup(ns, W, H, seed, **scene_kw):
    sc = make_scene(W, H, seed=seed, **scene_kw)
    normals = ns.depth2normal.depth2normal(sc["depths"], sc["src_intrinsics"])
    nerf = build_reference_nerf(ns)
    msd = make_mlp_state_dict()
    nerf.mlp_fine.load_state_dict(msd, strict=True)
    enc = nerf.encoder
    enc.depths, enc.depths_std, enc.normals = sc["depths"][None], sc["depths_std"][None], normals[None]
    enc.latent = sc["latent"][None]
    enc.nviews, enc.nobjects = sc["src_extrinsics"].shape[0], 1
    nerf.poses = sc["src_extrinsics"][None]
    nerf.c = sc["src_intrinsics"][None, :, :2, -1]
    nerf.focal = sc["src_intrinsics"][None][:, :, [0, 1], [0, 1]]
    nerf.image_shape = sc["image_shape"].clone()
    scene = O.Scene(latent=sc["latent"], depths=sc["depths"], depths_std=sc["depths_std"], normals=normals,
                    poses=sc["src_extrinsics"], focal=nerf.focal[0], c=nerf.c[0], image_shape=sc["image_shape"],
                    feature_padding=float(enc.feature_padding))
    w = O.MLPWeights.from_state_dict(msd)
    rays = ns.cam_geometry.gen_rays(sc["target_extrinsics"][None], sc["target_intrinsics"][None], W, H,
                                    torch.tensor([sc["znear"]]), torch.tensor([sc["zfar"]])).view(H * W, 8)
    return sc, nerf, scene, w, rays


def render_fixture(ns, name, W, H, seed, K, G, white, noise_seed, scene_kw, n_lattice=64, tie_likelihood_max=1e-4):
    n_cand = 1000
    sc, nerf, scene, w, rays = setup(ns, W, H, seed, **scene_kw)
    idx = lattice(W, H, n_lattice)
    rs = rays[idx].contiguous()
    NR = rs.shape[0]
    g = torch.Generator().manual_seed(noise_seed)
    ncz, ngz, nfz = torch.rand(NR, n_cand, generator=g), torch.randn(NR, G, generator=g), torch.rand(NR, K, generator=g)
    ren = ns.nerf_renderer.NeRFRendererDGS(n_samples=K, n_depth_candidates=n_cand, n_gaussian=G, white_bkgd=white)
    with inject_noise(ncz, ngz, nfz):
        out = ren.forward(nerf, rs[None], want_weights=True)
        z_ref = ren.fill_up_uniform_samples(
            ren.sample_depthguided(rs[None], nerf, n_samples=K, n_candidates=n_cand, n_gaussian=G), rs[None])[0]
    sub = slice(0, NR, 8)
    o = O.render(scene, w, rs[sub].contiguous(), K, n_cand, G, white, ncz[sub], ngz[sub], nfz[sub])
    report(f"{name} z", z_ref[sub], o["z"], exact=True)
    report(f"{name} rgb", out.fine.rgb[0][sub], o["rgb"])
    report(f"{name} depth", out.fine.depth[0][sub], o["depth"])
    report(f"{name} weights", out.fine.weights[0][sub], o["weights"])
    zc = O.sample_coarse(rs, n_cand, ncz)
    L, Oq = O.point_likelihood(scene, rs, zc)
    Ls = L.sort(dim=-1, descending=True).values
    # ties at the (K-G) cut-off make the reference's own pick implementation-defined (unstable sort);
    # none may involve a well-defined likelihood.  Ties among erf-saturation values (L = 3e-8, 6e-8: both erf values
    # within an ulp of +-1) do occur at the narrow Facescape sigmas; tests treat those rays like every other
    # saturation-class ray (tests/helpers.py::selection_diff).
    tie = (Ls[:, K - G - 1] == Ls[:, K - G]) & (Ls[:, K - G] > 0)
    ties, ties_sat = int((tie & (Ls[:, K - G] >= 1e-6)).sum()), int((tie & (Ls[:, K - G] < 1e-6)).sum())
    print(f"    rays with surface {(Oq != 0).any(-1).sum().item()}/{NR}, ties at the cut-off {ties} "
          f"(+{ties_sat} among saturation-class values < 1e-6), "
          f"rays with fewer than K-G positive candidates {((L > 0).sum(-1) < K - G).sum().item()}")
    tie_rays = (tie & (Ls[:, K - G] >= 1e-6)).nonzero().flatten()
    if ties:
        # exact ties between likelihoods above the saturation class: small values are quantised in steps of 3e-8 (half an
        # ulp of erf near 1), so a few coincide.  Which of two tied candidates the reference keeps is decided by torch's
        # unstable argsort on this host; the fixture lists those rays, tests do not require the same pick on them.
        print(f"    tie rays {tie_rays.tolist()} at likelihoods {[float(Ls[r, K - G]) for r in tie_rays]}")
    assert ties <= 8
    # exact ties at the cut-off only among small likelihoods: below 1e-4 for G9 / G10; G16 has one exact tie at 1.8e-3 (two candidates
    # mirrored about the surface) and says so in its call
    assert all(float(Ls[r, K - G]) < tie_likelihood_max for r in tie_rays), [float(Ls[r, K - G]) for r in tie_rays]
    np.savez_compressed(os.path.join(OUT, name + ".npz"), W=W, H=H, seed=seed, K=K, G=G, n_cand=n_cand,
                        white_bkgd=int(white), noise_seed=noise_seed, znear=sc["znear"], zfar=sc["zfar"],
                        ray_idx=idx.numpy(), rays=rs.numpy(), tie_rays=tie_rays.numpy(), in_sha=sha(ncz[:64], ngz[:64], nfz[:64]),
                        rgb=out.fine.rgb[0].numpy(), depth=out.fine.depth[0].numpy(), z=z_ref.numpy(),
                        weights_sum=out.fine.weights[0].sum(-1).numpy(), weights_sub=out.fine.weights[0][::16].numpy())


def helpers_fixture(ns):
    print("G11 helpers: torch_cmap, depth2normal with holes, gen_rays")
    g = torch.Generator().manual_seed(211)
    # torch_cmap: per-image min/max, explicit vmin/vmax, a value exactly at the maximum
    depth = torch.rand(2, 1, 45, 61, generator=g) * 0.9 + 0.5
    depth[0, 0, 3, 4] = depth[0].max()
    cm = {}
    for tag, (vmin, vmax) in dict(auto=(None, None), fixed=(0.25, 1.75), hi=(None, 2.0)).items():
        cm["cmap_" + tag] = ns.torch_helpers.torch_cmap(depth, vmin=vmin, vmax=vmax).numpy()      # (2,3,H,W) float64
    # depth2normal: isolated holes, hole on the border, a hole block, a zero column
    d2 = torch.rand(2, 1, 33, 47, generator=g) + 0.5
    d2[0, 0, 5, 7] = 0
    d2[0, 0, 0, 3] = 0
    d2[0, 0, 32, 46] = 0
    d2[1, 0, 10:14, 20:23] = 0
    d2[1, 0, :, 0] = 0
    K2 = torch.tensor([[[40.0, 0, 23.5], [0, 42.0, 16.5], [0, 0, 1]], [[38.0, 0, 20.0], [0, 38.0, 15.0], [0, 0, 1]]])
    n2 = ns.depth2normal.depth2normal(d2, K2)
    sc = make_scene(48, 40, seed=3, latent=False)
    n_scene = ns.depth2normal.depth2normal(sc["depths"], sc["src_intrinsics"])
    # gen_rays: three different cameras, non-square image, per-camera near/far
    E = torch.stack([sc["target_extrinsics"], sc["src_extrinsics"][1], sc["src_extrinsics"][3]])
    Km = torch.stack([sc["target_intrinsics"], sc["src_intrinsics"][1], sc["src_intrinsics"][3]])
    Wr, Hr = 37, 29
    zn, zf = torch.tensor([0.5, 0.6, 0.7]), torch.tensor([1.5, 1.6, 1.7])
    rays = ns.cam_geometry.gen_rays(E, Km, Wr, Hr, zn, zf)
    np.savez_compressed(os.path.join(OUT, "g11_helpers.npz"), cmap_depth=depth.numpy(), **cm,
                        d2n_depth=d2.numpy(), d2n_K=K2.numpy(), d2n_normals=n2.numpy(),
                        d2n_scene_seed=3, d2n_scene_normals=n_scene.numpy(),
                        rays_E=E.numpy(), rays_K=Km.numpy(), rays_W=Wr, rays_H=Hr, rays_near=zn.numpy(),
                        rays_far=zf.numpy(), rays=rays.numpy())
    print("    written")


def main():
    which = set(a.lower() for a in sys.argv[1:]) or {"g9", "g10", "g11", "g16"}
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count())
    ns = import_reference()
    with torch.no_grad():
        if "g11" in which:
            helpers_fixture(ns)
        if "g9" in which:
            print("G9 renderer.forward K=128/G=48 on 4096 rays of the 400x300 bench scene (a few minutes)")
            render_fixture(ns, "g9_render_K128", 400, 300, 0, 128, 48, False, 109, {})
        if "g10" in which:
            print("G10 renderer.forward K=192/G=72, white background, Facescape range, 4096 rays of 256x256")
            render_fixture(ns, "g10_render_cfg5", 256, 256, 0, 192, 72, True, 110,
                           dict(scale=1.75, znear=1.0, zfar=2.5, std_law="facescape"))
        if "g16" in which:
            print("G16 renderer.forward K=192/G=72, DTU sigma law and range, 2304 rays of the 400x300 bench scene")
            render_fixture(ns, "g16_render_K192_dtu", 400, 300, 0, 192, 72, False, 116, {}, n_lattice=48, tie_likelihood_max=2e-3)
    print("done")


if __name__ == "__main__":
    main()
