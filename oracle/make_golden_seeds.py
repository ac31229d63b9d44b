"""G17: the reference renderer's OWN seed-to-seed distance on the render fixtures (build container only; test infrastructure).

    python oracle/make_golden_seeds.py [g9] [g10] [g16]          # default: all three

The HIP sampler picks a different but equally valid sample set on rays whose candidates straddle the selection cut-off within erf
round-off (tests/helpers.py::selection_diff): such a ray is rendered from other stratified samples, which is what
another noise seed does to EVERY ray.  This script renders the scenes of G9 / G10 / G16 again with the imported reference
(nerf_renderer.py:399-430) and two other noise seeds -- the three draws of nerf_renderer.py:57, :188, :390 -- and stores those images.
The GPU tests then have the denominator: the reference's seed-to-seed PSNR and per-ray spread next to the HIP-vs-reference distance
on the differing rays, and a bias test (mean colour difference over the differing rays against the seed-to-seed standard error).
No DTU data or checkpoint exists here, so this is the available stand-in for "PSNR within 0.05 dB on DTU val" (BASELINE.json).
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle.ref_import import import_reference                           # noqa: E402
from oracle.make_golden import inject_noise, OUT                         # noqa: E402
from oracle.make_golden_r2 import setup, lattice                         # noqa: E402

CASES = {   # name: (W, H, scene seed, K, G, white, noise seed of the fixture, scene kwargs, lattice) -- as make_golden_r2.py
    "g9": ("g9_render_K128", 400, 300, 0, 128, 48, False, 109, {}, 64),
    "g10": ("g10_render_cfg5", 256, 256, 0, 192, 72, True, 110, dict(scale=1.75, znear=1.0, zfar=2.5, std_law="facescape"), 64),
    "g16": ("g16_render_K192_dtu", 400, 300, 0, 192, 72, False, 116, {}, 48),
}
EXTRA_SEEDS = (1000, 2000)       # added to the fixture's noise seed
# `python oracle/make_golden_seeds.py ensemble [g9] [g10] [g16]`: six MORE seeds per fixture -> tests/golden/g18_seed_ensemble.npz.  With the two
# above that is an 8-member ensemble of the reference's own renders whose mean is the "ground truth" of a PSNR-vs-ground-truth comparison: the
# stand-in for north_star's "PSNR within 0.05 dB of reference on DTU val" (the GPU test compares PSNR(HIP image, ensemble mean) with
# PSNR(reference's fixture image, ensemble mean)).
ENSEMBLE_SEEDS = (3000, 4000, 5000, 6000, 7000, 8000)
# `python oracle/make_golden_seeds.py more g16`: EIGHT more members (round 5: a 16-member mean for G16, whose 8-member "ground truth" left
# the 0.05 dB margin of the test no head-room at a 19.5 dB seed-to-seed spread) -> `<key>_rgb_more` in the same file, the first six untouched
ENSEMBLE_MORE = (9000, 10000, 11000, 12000, 13000, 14000, 15000, 16000)


def psnr(a, b):
    return float(-10.0 * torch.log10(((a - b) ** 2).mean()))


def main():
    args = [a.lower() for a in sys.argv[1:]]
    ensemble = "ensemble" in args or "more" in args
    more = "more" in args
    which = [a for a in args if a not in ("ensemble", "more")] or list(CASES)
    torch.set_num_threads(os.cpu_count())
    ns = import_reference()
    if ensemble:
        return make_ensemble(ns, which, ENSEMBLE_MORE if more else ENSEMBLE_SEEDS, "_rgb_more" if more else "_rgb")
    path = os.path.join(OUT, "g17_seed_to_seed.npz")
    store = dict(np.load(path)) if os.path.exists(path) else {}
    with torch.no_grad():
        for key in which:
            name, W, H, seed, K, G, white, noise_seed, scene_kw, n_lat = CASES[key]
            fix = np.load(os.path.join(OUT, name + ".npz"))
            sc, nerf, scene, w, rays = setup(ns, W, H, seed, **scene_kw)
            rs = rays[lattice(W, H, n_lat)].contiguous()
            assert np.array_equal(rs.numpy(), fix["rays"]), "ray lattice differs from the fixture's"
            NR, n_cand = rs.shape[0], 1000
            ren = ns.nerf_renderer.NeRFRendererDGS(n_samples=K, n_depth_candidates=n_cand, n_gaussian=G, white_bkgd=white)
            ref_rgb, ref_d = torch.from_numpy(fix["rgb"]), torch.from_numpy(fix["depth"])
            imgs = []
            for s in EXTRA_SEEDS:
                g = torch.Generator().manual_seed(noise_seed + s)
                nz = (torch.rand(NR, n_cand, generator=g), torch.randn(NR, G, generator=g), torch.rand(NR, K, generator=g))
                with inject_noise(*nz):
                    out = ren.forward(nerf, rs[None])
                imgs.append((out.fine.rgb[0].clone(), out.fine.depth[0].clone()))
                print(f"{name} seed +{s}: PSNR against the fixture's image {psnr(out.fine.rgb[0], ref_rgb):.2f} dB, "
                      f"per-ray max colour difference: max {float((out.fine.rgb[0] - ref_rgb).abs().max()):.3f}, "
                      f"median {float((out.fine.rgb[0] - ref_rgb).abs().max(-1).values.median()):.2e}", flush=True)
            print(f"{name} seed +{EXTRA_SEEDS[0]} vs +{EXTRA_SEEDS[1]}: {psnr(imgs[0][0], imgs[1][0]):.2f} dB")
            for i, (rgb, d) in enumerate(imgs):
                store[f"{key}_rgb_s{i + 1}"] = rgb.numpy()
                store[f"{key}_depth_s{i + 1}"] = d.numpy()
            store[f"{key}_psnr_s1_vs_fixture"] = np.float64(psnr(imgs[0][0], ref_rgb))
            store[f"{key}_psnr_s2_vs_fixture"] = np.float64(psnr(imgs[1][0], ref_rgb))
            store[f"{key}_psnr_s1_vs_s2"] = np.float64(psnr(imgs[0][0], imgs[1][0]))
            store[f"{key}_noise_seeds"] = np.array([noise_seed + s for s in EXTRA_SEEDS])
            np.savez_compressed(path, **store)
    print("done", path)


def make_ensemble(ns, which, seeds=ENSEMBLE_SEEDS, suffix="_rgb"):
    path = os.path.join(OUT, "g18_seed_ensemble.npz")
    store = dict(np.load(path)) if os.path.exists(path) else {}
    with torch.no_grad():
        for key in which:
            name, W, H, seed, K, G, white, noise_seed, scene_kw, n_lat = CASES[key]
            fix = np.load(os.path.join(OUT, name + ".npz"))
            sc, nerf, scene, w, rays = setup(ns, W, H, seed, **scene_kw)
            rs = rays[lattice(W, H, n_lat)].contiguous()
            assert np.array_equal(rs.numpy(), fix["rays"]), "ray lattice differs from the fixture's"
            NR, n_cand = rs.shape[0], 1000
            ren = ns.nerf_renderer.NeRFRendererDGS(n_samples=K, n_depth_candidates=n_cand, n_gaussian=G, white_bkgd=white)
            imgs = []
            for s in seeds:
                g = torch.Generator().manual_seed(noise_seed + s)
                nz = (torch.rand(NR, n_cand, generator=g), torch.randn(NR, G, generator=g), torch.rand(NR, K, generator=g))
                with inject_noise(*nz):
                    out = ren.forward(nerf, rs[None])
                imgs.append(out.fine.rgb[0].clone())
                print(f"{name} ensemble seed +{s}: PSNR against the fixture's image {psnr(imgs[-1], torch.from_numpy(fix['rgb'])):.2f} dB", flush=True)
            store[f"{key}{suffix}"] = torch.stack(imgs).numpy().astype(np.float32)          # (6 | 8, NR, 3)
            store[f"{key}_noise_seeds{suffix[4:]}"] = np.array([noise_seed + s for s in seeds])
            np.savez_compressed(path, **store)
    print("done", path)


if __name__ == "__main__":
    main()
