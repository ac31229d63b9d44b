"""GPU parity AT THE MAP SIZES of BASELINE configs[2]/[3] (800x600, K = 128, latent 364 x 464, 4.2 GB of projected maps) and
configs[4] (1024 x 1024, K = 192, Facescape range, white background, latent 576 x 576, 8.2 GB of projected maps + the fp16 copy):
HIP against the CPU oracle on the SAME rays with explicit noise.

The full-frame tests of test_hip_parity.py state size-independent properties (sharding / batching invariance, determinism, ranges)
-- every one of them blind to an addressing, hoist or fp16-map error that is the same in both renders being compared.  Here
the oracle (pinned bit-exact against the imported reference, oracle/make_golden*.py) evaluates 512 rays of the very scenes
bench.py renders: the four corner windows of the frame, a centre window, and a wide-angle window whose border rays project
OUTSIDE EVERY SOURCE VIEW (border-clamped latent / depth taps, the zero ring of the padded std map, zero normals).
Reference: image_encoder.py:112-123 (uv scaling by the padded map size), pixelnerf.py:105-116, nerf_renderer.py:65-190,
create_prediction_folder.py:44-47.

Each window is a small target image of its own (principal point shifted into the frame: the rays ARE the frame's rays of those
pixels), so the same rays also go through the drop-in modules and the image harness (`predict_image`: device ray generation,
ragged ray batches, `renderer.forward` of src.models)."""
import numpy as np
import pytest
import torch

from oracle import diner_oracle as O
from tests.helpers import oracle_setup, max_norm_rel, SAT_L

pytestmark = pytest.mark.gpu
TOL = 1e-4

CONFIGS = {
    # name: (W, H, K, G, white, scene kwargs, max share of rays whose pick set may differ (erf round-off classes A / B of
    #        test_render_at_metric_sample_counts: 0.27 % at K = 128, 24 % at K = 192 with the narrow Facescape sigmas))
    "cfg3_800x600_K128": (800, 600, 128, 48, False, dict(), 0.03),
    "cfg5_1024_K192": (1024, 1024, 192, 72, True, dict(scale=1.75, znear=1.0, zfar=2.5, std_law="facescape"), 0.40),
}
N_CAND = 1000


def _windows(W, H, Kt):
    """-> list of (label, intrinsics (3,3), w, h): target 'images' whose pixels are pixels of the W x H frame (shifted principal
    point), + one wide-angle camera (focal 0.3 x its width: +-59 degrees) whose border rays leave every source view."""
    def crop(x0, y0, w, h):
        Kc = Kt.clone()
        Kc[0, 2] -= x0
        Kc[1, 2] -= y0
        return Kc, w, h
    wins = [("corner00",) + crop(0, 0, 8, 8), ("corner01",) + crop(W - 8, 0, 8, 8), ("corner10",) + crop(0, H - 8, 8, 8),
            ("corner11",) + crop(W - 8, H - 8, 8, 8), ("centre",) + crop(W // 2 - 8, H // 2 - 4, 16, 8)]
    Kw = torch.tensor([[0.3 * 16, 0.0, 8.0], [0.0, 0.3 * 16, 4.0], [0.0, 0.0, 1.0]])
    wins.append(("wide", Kw, 16, 8))
    return wins


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from diner_amd import ops as _ops
    return _ops


def _errs(rgb, depth, ref_rgb, ref_d):
    return ((rgb.cpu() - ref_rgb).abs().max(-1).values / ref_rgb.abs().max(), (depth.cpu() - ref_d).abs() / ref_d.abs().max())


@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_bench_scene_against_oracle(ops, name):
    from diner_amd import noise
    from diner_amd.render import predict_image
    from diner_amd.synthetic import build_modules
    W, H, K, G, white, kw, max_diff_share = CONFIGS[name]
    sc, scene, w, msd, _ = oracle_setup(W, H, 0, **kw)                    # bench.py's scene of this configuration (seed 0)
    nerf, R = build_modules(sc, msd, "cuda", normals=sc["normals"])
    ren = R(n_samples=40, n_depth_candidates=N_CAND, n_gaussian=15, white_bkgd=white)
    ren.n_samples, ren.n_gaussian = K, int(15 * K / 40)                    # as create_prediction_folder.py:44-47 does
    assert ren.n_gaussian == G
    hs, hm = nerf.hip_scene(0), nerf.hip_mlp()
    E = sc["target_extrinsics"][None].cuda()
    wins = _windows(W, H, sc["target_intrinsics"])
    gen = torch.Generator().manual_seed(20250 + K)
    rays_l, noise_l, img_l = [], [], []
    for label, Kc, wd, ht in wins:
        n = wd * ht
        r = ops.gen_rays(E, Kc[None], wd, ht, sc["znear"], sc["zfar"], "cuda")[0]          # the rays predict_image generates
        nz = (torch.rand(n, N_CAND, generator=gen), torch.randn(n, G, generator=gen), torch.rand(n, K, generator=gen))
        # (ii) the image harness through the drop-in modules: ragged ray batches, injected noise sliced per batch
        with noise.inject(*(t.cuda()[None] for t in nz)):
            rgb_i, dep_i = predict_image(nerf, ren, E, Kc[None].cuda(), wd, ht, sc["znear"], sc["zfar"], ray_batch_size=50)
        assert rgb_i.shape == (1, 3, ht, wd) and dep_i.shape == (1, 1, ht, wd)
        img_l.append(torch.cat((rgb_i[0].permute(1, 2, 0).reshape(n, 3), dep_i[0, 0].reshape(n, 1)), -1).cpu())
        rays_l.append(r.cpu())
        noise_l.append(nz)
    rays = torch.cat(rays_l).contiguous()
    nc, ng, nf = (torch.cat([nz[i] for nz in noise_l]).contiguous() for i in range(3))
    img = torch.cat(img_l)
    NR = rays.shape[0]
    assert NR == 512
    # the frame's own corner pixels are among the rays (pixel centres at +0.5, cam_geometry.py:28-34)
    full = O.gen_rays(sc["target_extrinsics"], sc["target_intrinsics"], W, H, sc["znear"], sc["zfar"])
    for wi, pix in ((0, 0), (1, W - 1), (2, (H - 1) * W), (3, H * W - 1)):
        local = {0: 0, 1: 7, 2: 56, 3: 63}[wi]
        assert (rays[64 * wi + local] - full[pix]).abs().max().item() <= 5e-7
    del full
    # ---- the oracle on these rays
    ref = O.render(scene, w, rays, K, N_CAND, G, white, nc, ng, nf)
    ref_rgb, ref_d, ref_z = ref["rgb"], ref["depth"], ref["z"]
    # coverage: sample points that project outside every source view (|u| or |v| > 1 in all four), and outside the 100-px std ring
    pts = (rays[:, None, :3] + ref_z[..., None] * rays[:, None, 3:6]).reshape(-1, 3)
    uv = O.project_uv(scene, O.world_to_cam(scene, pts))
    outside_all = ((uv.abs() > 1).any(-1)).all(0).view(NR, K)
    ring = 1 + 200.0 / min(W, H)
    beyond_ring = ((uv.abs() > ring).any(-1)).all(0).view(NR, K)
    n_out_rays, n_ring_rays = int(outside_all.all(-1).sum()), int(beyond_ring.all(-1).sum())
    print(f"{name}: {int(outside_all.sum())} of {NR * K} sample points outside every source view ({n_out_rays} rays entirely, "
          f"{n_ring_rays} rays entirely beyond the std padding ring); feature map {tuple(sc['latent'].shape[-2:])}")
    assert n_out_rays >= 16 and int(outside_all.sum()) >= 0.05 * NR * K

    rc = rays.cuda()
    zc = O.sample_coarse(rays, N_CAND, nc)
    L, Occ = O.point_likelihood(scene, rays, zc)
    for mode in ("f16x3", "fp32"):
        # (1) field + compositor on the ORACLE's sample positions: every ray
        wts, rgb, depth = ops.render(hs, hm, rc, ref_z.cuda(), white, want_weights=True, precision=mode)
        e_rgb, e_d = _errs(rgb, depth, ref_rgb, ref_d)
        print(f"{name} [{mode}] oracle z: rgb {e_rgb.max().item():.2e} depth {e_d.max().item():.2e} (all {NR} rays), "
              f"weights {max_norm_rel(wts.cpu(), ref['weights']):.2e}")
        assert e_rgb.max().item() < TOL and e_d.max().item() < TOL
        assert max_norm_rel(wts.cpu(), ref["weights"]) < TOL
        assert max_norm_rel(ops.field_from_rays(hs, hm, rc, ref_z.cuda(), precision=mode).cpu(), ref["field"]) < 2e-5
    # (2) the sampler at this map size: picks, gaussian slots, fill
    z, zu = ops.sample_depthguided(hs, rc, K, N_CAND, G, 0.05, noise=(nc.cuda(), ng.cuda(), nf.cuda()), want_unfilled=True)
    zh, zuh = z.cpu(), zu.cpu()
    same = torch.isclose(zh, ref_z, rtol=3e-6, atol=1e-7).all(-1)
    diff = (~same).nonzero().flatten()
    n_a = n_b = 0
    for r in diff.tolist():
        only = set(ref_z[r].tolist()) ^ set(zh[r].tolist())
        cand = [int((zc[r] == zz).nonzero().flatten()[0]) for zz in only if (zc[r] == zz).any()]
        if cand:                                   # class A: candidates within erf round-off of the ray's cut-off likelihood
            n_a += 1
            Ls = L[r].sort(descending=True).values
            cut = float(Ls[K - G - 1])
            worst = max(abs(float(L[r, c]) - cut) for c in cand)
            assert worst < SAT_L, f"ray {r}: pick differs on a candidate {worst:.1e} away from the cut-off likelihood"
        else:                                      # class B: same picks, gaussian fit on round-off residue
            n_b += 1
            assert 0 < float(Occ[r].sum()) < 1e-2, f"ray {r}: same picks, well-conditioned gaussian fit, different samples"
    assert len(diff) <= max_diff_share * NR, (len(diff), n_a, n_b)
    assert bool(same[outside_all.all(-1)].all()), "a ray outside every source view has no likelihood: its samples are the fill alone"
    if len(diff):
        refill = O.fill_up_uniform_samples(zuh[diff], rays[diff].contiguous(), nf[diff])
        assert torch.equal(refill, zh[diff]), "the reference's fill of the HIP pick set is not the HIP sample set"
    # (3) renderer on its own samples, ops level (both parity-grade modes) and through predict_image / src.models
    o_rgb = o_d = None
    if len(diff):                                  # oracle field + compositor AT THE HIP SAMPLES on the rays whose set differs
        _, o_rgb, o_d, _ = O.composite(scene, w, rays[diff].contiguous(), zh[diff].contiguous(), white)
    outs = {}
    for mode in ("f16x3", "fp32"):
        _, rgb, depth = ops.render(hs, hm, rc, z, white, precision=mode)
        outs[mode] = (rgb.cpu(), depth.cpu())
    outs["predict_image"] = (img[:, :3], img[:, 3])
    default = {ops.PRECISION_FP32: "fp32", ops.PRECISION_F16X3: "f16x3"}[ops.get_precision()]      # what the modules ran in
    assert torch.equal(outs["predict_image"][0], outs[default][0]), "module path and ops path differ on the same rays"
    assert torch.equal(outs["predict_image"][1], outs[default][1])
    for label, (rgb, depth) in outs.items():
        e_rgb, e_d = _errs(rgb, depth, ref_rgb, ref_d)
        hot = (same & ((e_rgb >= TOL) | (e_d >= TOL))).nonzero().flatten()      # gaussian-sample shifts amplified by the depth code
        assert len(hot) <= 3, (label, len(hot))
        cool = same.clone()
        cool[hot] = False
        assert e_rgb[cool].max().item() < TOL and e_d[cool].max().item() < TOL, label
        msg = f"{name} [{label}] own samples: {int(cool.sum())} rays with the oracle's sample set rgb {e_rgb[cool].max().item():.2e} depth {e_d[cool].max().item():.2e}"
        if len(diff):
            h_rgb, h_d = _errs(rgb[diff], depth[diff], o_rgb, o_d)
            assert h_rgb.max().item() < TOL and h_d.max().item() < TOL, label
            msg += f"; {len(diff)} rays with another pick set (A {n_a}, B {n_b}) against the oracle at the HIP samples rgb {h_rgb.max().item():.1e} depth {h_d.max().item():.1e}"
        print(msg)
    # (4) the arithmetic configs[4] names (plain fp16 operands, taps from the fp16 copy of the projected maps): a PSNR statement
    if K == 192:
        _, rgb16, d16 = ops.render(hs, hm, rc, ref_z.cuda(), white, precision="f16")
        mse = (rgb16.cpu() - ref_rgb).square().mean().item()
        psnr = 10 * np.log10(1.0 / max(mse, 1e-30))
        e_rgb, e_d = _errs(rgb16, d16, ref_rgb, ref_d)
        print(f"{name} [f16] oracle z: PSNR {psnr:.1f} dB, rgb {e_rgb.max().item():.2e} depth {e_d.max().item():.2e}")
        assert psnr >= 70.0 and e_rgb.max().item() < 5e-3 and e_d.max().item() < 5e-3
    assert hm.fallback_launches() == 0


@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_gathers_on_bench_size_maps(ops, name):
    """The four gathers (image_encoder.py:97-223) on the maps of the bench scenes: uv on a grid that includes exactly +-1, the pixel
    edges, half-pixel positions, points up to 100 px outside (padded std ring) and far outside -- nearest taps bit-exact, the bilinear
    512-channel lookup to 5e-6, at Wf x Hf = 464 x 364 and 576 x 576 (tap offsets of the last texel rows: (Hf Wf - 1) 512 floats x 4 views
    = 2.7 GB into the buffer at 1024 x 1024)."""
    W, H, K, G, white, kw, _ = CONFIGS[name]
    sc, scene, w, msd, _ = oracle_setup(W, H, 0, **kw)
    hs = ops.HipScene(sc["latent"].cuda(), sc["depths"].cuda(), sc["depths_std"].cuda(), sc["normals"].cuda(), sc["src_extrinsics"],
                      sc["src_intrinsics"][:, [0, 1], [0, 1]], sc["src_intrinsics"][:, :2, -1], sc["image_shape"], sc["feature_padding"])
    g = torch.Generator().manual_seed(7)
    edge = torch.tensor([-1.0, 1.0, -1.0 + 1.0 / W, 1.0 - 1.0 / W, -1.0 - 1.0 / W, 1.0 + 1.0 / W, 0.0, 1.0 - 2.0 / W, -1.0 + 2.0 / W,
                         1.0 + 199.0 / W, -1.0 - 199.0 / W, 1.0 + 201.0 / W, -1.0 - 201.0 / W, 3.0, -3.0])
    uu, vv = torch.meshgrid(edge, edge, indexing="ij")
    grid = torch.stack((uu, vv), -1).reshape(-1, 2)
    rnd = torch.rand(1024, 2, generator=g) * 2.6 - 1.3
    uv1 = torch.cat((grid, rnd))
    uv = torch.stack([uv1 * (1 - 0.01 * v) for v in range(4)]).contiguous()          # (NV, N, 2), another set per view
    uvc = uv.cuda()
    assert torch.equal(ops.index(hs, ops.INDEX_DEPTH, uvc).cpu(), O.index_depth(scene, uv))
    assert torch.equal(ops.index(hs, ops.INDEX_DEPTH_STD, uvc).cpu(), O.index_depth_std(scene, uv))
    assert torch.equal(ops.index(hs, ops.INDEX_NORMAL, uvc).cpu(), O.index_normal(scene, uv))
    lat, want = ops.index(hs, ops.INDEX_LATENT, uvc).cpu(), O.index_latent(scene, uv)
    rel = max_norm_rel(lat, want)
    print(f"{name}: latent {tuple(sc['latent'].shape)} bilinear max-norm-rel {rel:.2e} on {uv.shape[1]} uv per view (|uv| up to 3)")
    assert rel < 5e-6
    # the last texel of the last view (largest tap offset) is reachable and correct
    far = torch.tensor([[[3.0, 3.0]]]).repeat(4, 1, 1)
    assert torch.equal(ops.index(hs, ops.INDEX_LATENT, far.cuda()).cpu()[:, :, 0], sc["latent"][:, :, -1, -1])
