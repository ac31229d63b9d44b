"""GPU parity: every HIP stage (through the C ABI, via diner_amd.ops) against the golden vectors generated
from the imported reference and against the CPU oracle on the same seeded inputs.

Tolerances: north_star asks for 1e-4 relative fp32 on the renderer outputs; the fp32 noise floor of the
reference path itself is ~5e-6.  Index/selection work is compared exactly.
"""
import numpy as np
import pytest
import torch

from oracle import diner_oracle as O
from tests.helpers import load, oracle_setup, max_norm_rel, sha, selection_diff, SAT_L

pytestmark = pytest.mark.gpu

TOL = 1e-4          # north_star tolerance on rendered outputs
TOL_STAGE = 2e-5    # what the individual fp32 stages are held to


def T(a):
    return torch.from_numpy(np.asarray(a))


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from diner_amd import ops as _ops
    return _ops


@pytest.fixture(params=["f16x3", "fp32"])
def precision(request, ops):
    """Both parity-grade arithmetic modes of the MLP GEMMs are held to the same tolerances (the mode is a per-call argument
    of the C ABI; set_precision only changes the default of this Python host)."""
    prev = ops.get_precision()
    ops.set_precision(request.param)
    yield request.param
    ops.set_precision(prev)


def hip_scene(ops, sc):
    K = sc["src_intrinsics"]
    return ops.HipScene(sc["latent"].cuda(), sc["depths"].cuda(), sc["depths_std"].cuda(), sc["normals"].cuda(),
                        sc["src_extrinsics"], K[:, [0, 1], [0, 1]], K[:, :2, -1], sc["image_shape"],
                        sc["feature_padding"])


def hip_mlp(ops, msd):
    return ops.HipMlp({k: v.cuda() for k, v in msd.items()})


def test_posenc(ops):
    g = load("g1_posenc.npz")
    for x, y in ((g["x3"], g["y3"]), (g["x1"], g["y1"])):
        out = ops.posenc(T(x).cuda(), 6, 6.28, True).cpu()
        err = (out - T(y)).abs().max().item()
        print(f"posenc d={x.shape[-1]} max-abs err {err:.3e}")
        assert err < 2e-6
        assert torch.equal(out[:, :x.shape[-1]], T(x))


def test_gathers(ops):
    g = load("g2_gathers.npz")
    sc, scene, w, msd, rays = oracle_setup(int(g["W"]), int(g["H"]), int(g["seed"]), bg_std_zero=True)
    hs = hip_scene(ops, sc)
    uv = T(g["uv"]).cuda()
    d = ops.index(hs, ops.INDEX_DEPTH, uv).cpu()
    s = ops.index(hs, ops.INDEX_DEPTH_STD, uv).cpu()
    n = ops.index(hs, ops.INDEX_NORMAL, uv).cpu()
    lat = ops.index(hs, ops.INDEX_LATENT, uv).cpu()
    assert torch.equal(d, T(g["depth"])), "nearest/border depth taps must be bit-exact"
    assert torch.equal(s, T(g["std"])), "exponentially padded std taps must be bit-exact"
    assert torch.equal(n, T(g["normal"])), "nearest/zeros normal taps must be bit-exact"
    rel = max_norm_rel(lat[:, ::16], g["latent_sub"])
    print(f"latent bilinear max-norm-rel {rel:.3e}")
    assert rel < 5e-6          # 4-tap blend, association differs from ATen's
    assert max_norm_rel(lat, O.index_latent(scene, uv.cpu())) < 5e-6


def _sampler_case(K):
    g = load(f"g3_sampler_K{K}.npz")
    sc, scene, w, msd, rays = oracle_setup(int(g["W"]), int(g["H"]), int(g["seed"]))
    gen = torch.Generator().manual_seed(103)
    sel = torch.randperm(int(g["W"]) * int(g["H"]), generator=gen)[:512].sort().values
    rs = T(g["rays"])
    noises = {}
    for (KK, G) in [(64, 24), (128, 48)]:
        noises[KK] = (torch.rand(512, 1000, generator=gen), torch.randn(512, G, generator=gen),
                      torch.rand(512, KK, generator=gen))
    assert sha(*noises[K]) == str(g["in_sha"]), "seeded noise not reproducible on this host"
    return g, sc, scene, rs, noises[K]


@pytest.mark.parametrize("K", [64, 128])
def test_sampler_and_fill(ops, K):
    g, sc, scene, rs, (nc, ng, nf) = _sampler_case(K)
    G = int(g["G"])
    hs = hip_scene(ops, sc)
    z, zu = ops.sample_depthguided(hs, rs.cuda(), K, 1000, G, 0.05, noise=(nc.cuda(), ng.cuda(), nf.cuda()),
                                   want_unfilled=True)
    z, zu = z.cpu(), zu.cpu()
    ref_u = T(g["z_unfilled"]).sort(-1).values
    got_u = zu.sort(-1).values
    ref_z = T(g["z"])
    # depth-guided picks are candidate depths -> bit-exact; gaussian samples go through a reduction whose
    # association differs (wave tree vs torch's cascade) -> 3e-6 relative
    zc = O.sample_coarse(rs, 1000, nc)
    L, _ = O.point_likelihood(scene, rs, zc)
    bad, worst = selection_diff(ref_u, got_u, L, zc, K - G)
    exact_rows = (got_u == ref_u).all(-1).sum().item()
    print(f"K={K}: rays bit-exact {exact_rows}/512; rays with a different pick set: {len(bad)} "
          f"(largest likelihood distance from the cut-off {worst:.2e}); elements bit-exact {(got_u == ref_u).float().mean().item():.5f}")
    assert worst < SAT_L, "selection differs on a candidate whose likelihood is well separated from the cut-off"
    assert len(bad) <= 0.02 * 512
    assert (z[:, 1:] >= z[:, :-1]).all()
    good = torch.ones(512, dtype=torch.bool)
    good[bad] = False
    closez = torch.isclose(z[good], ref_z[good], rtol=3e-6, atol=1e-7)
    print(f"K={K}: filled z within 3e-6 on the {int(good.sum())} agreeing rays: {closez.float().mean().item():.6f}")
    assert closez.all()


def test_fill_handmade(ops):
    g = load("g4_fill.npz")
    z = ops.fill_uniform(T(g["z_in"]).cuda(), T(g["rays"]).cuda(), T(g["noise"]).cuda()).cpu()
    assert torch.equal(z, T(g["z_out"]))


def test_mlp_forward(ops):
    g = load("g5_mlp.npz")
    sc, scene, w, msd, rays = oracle_setup(16, 16, 0)
    zx = torch.randn(4, 300, 567, generator=torch.Generator().manual_seed(105))
    y = ops.mlp_forward(hip_mlp(ops, msd), zx.cuda()).cpu()
    rel = max_norm_rel(y, g["y"])
    print(f"ResnetFC (4,300,567) max-norm-rel vs reference {rel:.3e}")
    assert rel < TOL_STAGE


def test_pixelnerf_forward(ops, precision):
    g = load("g6_pixelnerf.npz")
    sc, scene, w, msd, rays = oracle_setup(int(g["W"]), int(g["H"]), int(g["seed"]))
    hs, hm = hip_scene(ops, sc), hip_mlp(ops, msd)
    out = ops.field_from_points(hs, hm, T(g["pts"]).cuda(), T(g["dirs"]).cuda()).cpu()
    rel = max_norm_rel(out, g["out"])
    print(f"PixelNeRF.forward (512 pts) [{precision}] max-norm-rel vs reference {rel:.3e}")
    assert rel < TOL_STAGE


def test_composite_and_render(ops, precision):
    g = load("g7_composite.npz")
    sc, scene, w, msd, rays = oracle_setup(int(g["W"]), int(g["H"]), int(g["seed"]))
    assert sha(sc["latent"], sc["depths"], sc["depths_std"], scene.normals, sc["src_extrinsics"],
               *[v for k, v in sorted(msd.items())]) == str(g["scene_sha"]), "seeded scene not reproducible on this host"
    hs, hm = hip_scene(ops, sc), hip_mlp(ops, msd)
    r7, z7, field = T(g["rays"]).cuda(), T(g["z"]).cuda(), T(g["field"]).cuda()
    for wb in (0, 1):
        wts, rgb, depth = ops.composite(field, z7, r7, bool(wb))
        for name, got in (("weights", wts), ("rgb", rgb), ("depth", depth)):
            rel = max_norm_rel(got.cpu(), g[f"{name}_{wb}"])
            print(f"composite white={wb} {name}: {rel:.3e}")
            assert rel < TOL_STAGE
        wts, rgb, depth = ops.render(hs, hm, r7, z7, bool(wb), want_weights=True)
        for name, got in (("weights", wts), ("rgb", rgb), ("depth", depth)):
            rel = max_norm_rel(got.cpu(), g[f"{name}_{wb}"])
            print(f"render    white={wb} {name}: {rel:.3e}")
            assert rel < TOL


def test_render_cfg1_end_to_end(ops, precision):
    """BASELINE.json configs[0]: 64x64 target, 64 samples/ray, 4 source views, against the reference's output.

    Two statements: (1) with the reference's own sample positions the renderer matches on EVERY ray;
    (2) with the HIP sampler the image matches on every ray whose sample set agrees with the reference's, and
    the rays that do not agree are the erf-saturation class of test_sampler_and_fill (a fraction of a percent)."""
    g = load("g8_render_cfg1.npz")
    W, H, K, G, n_cand = (int(g[k]) for k in ("W", "H", "K", "G", "n_cand"))
    sc, scene, w, msd, rays = oracle_setup(W, H, int(g["seed"]))
    gen = torch.Generator().manual_seed(108)
    nc = torch.rand(W * H, n_cand, generator=gen)
    ng = torch.randn(W * H, G, generator=gen)
    nf = torch.rand(W * H, K, generator=gen)
    assert sha(nc[:64], ng[:64], nf[:64]) == str(g["in_sha"]), "seeded noise not reproducible on this host"
    hs, hm = hip_scene(ops, sc), hip_mlp(ops, msd)
    rays = T(g["rays"])
    rc = rays.cuda()
    ref_rgb, ref_d, ref_z = T(g["rgb"]), T(g["depth"]), T(g["z"])

    def errs(rgb, depth):
        e_rgb = (rgb.cpu() - ref_rgb).abs().max(-1).values / ref_rgb.abs().max()
        e_d = (depth.cpu() - ref_d).abs() / ref_d.abs().max()
        return e_rgb, e_d

    # (1) reference z -> HIP field + compositor
    wts, rgb, depth = ops.render(hs, hm, rc, ref_z.cuda(), False, want_weights=True)
    e_rgb, e_d = errs(rgb, depth)
    print(f"e2e cfg1 [{precision}], reference z : rgb max-norm-rel {e_rgb.max().item():.3e}, depth {e_d.max().item():.3e}")
    assert e_rgb.max().item() < TOL and e_d.max().item() < TOL
    np.testing.assert_allclose(wts.cpu().sum(-1).numpy(), g["weights_sum"], atol=2e-5)
    # (2) HIP sampler -> HIP field + compositor
    z = ops.sample_depthguided(hs, rc, K, n_cand, G, 0.05, noise=(nc.cuda(), ng.cuda(), nf.cuda()))
    same = torch.isclose(z.cpu(), ref_z, rtol=3e-6, atol=1e-7).all(-1)
    wts, rgb, depth = ops.render(hs, hm, rc, z, False, want_weights=True)
    e_rgb, e_d = errs(rgb, depth)
    n_diff = int((~same).sum())
    print(f"e2e cfg1, HIP sampler : {n_diff}/{W * H} rays with a different sample set; on the others rgb "
          f"{e_rgb[same].max().item():.3e}, depth {e_d[same].max().item():.3e}; on all rays rgb "
          f"{e_rgb.max().item():.3e}, depth {e_d.max().item():.3e}")
    assert n_diff <= 0.005 * W * H
    assert e_rgb[same].max().item() < TOL and e_d[same].max().item() < TOL
    # the metric's "PSNR vs ref": the whole 64x64 image (all rays, including the handful with a different sample set)
    # against the reference's image, colours in [0, 1]
    mse = (rgb.cpu() - ref_rgb).square().mean().item()
    mse_same = (rgb.cpu() - ref_rgb)[same].square().mean().item()
    psnr, psnr_same = (10 * np.log10(1.0 / max(m, 1e-30)) for m in (mse, mse_same))
    print(f"e2e cfg1 [{precision}]: PSNR of the HIP image against the reference's image {psnr:.1f} dB "
          f"({psnr_same:.1f} dB without the {n_diff} erf-saturation rays)")
    # an image 55 dB from the reference's moves a ~30 dB PSNR-vs-ground-truth by < 0.01 dB (the metric allows 0.05)
    assert psnr > 55.0 and psnr_same > 90.0


# margin of |PSNR(HIP, ground truth) - PSNR(reference's image, ground truth)| per fixture, see statement (5) of the render test: north_star's 0.05 dB
PSNR_VS_GT_MARGIN_DB = {"g9": 0.05, "g10": 0.05, "g16": 0.05}

RENDER_FIXTURES = {
    # name: (scene kwargs of diner_amd.synthetic.make_scene,
    #        max number of class-A rays, max number of class-B rays (erf round-off classes, see the test), each pinned a few per cent
    #        above what the kernel measures on MI355X (deterministic: the HIP picks do not depend on the box),
    #        PSNR floor of the whole image against the reference's image, all rays included)
    # measured on MI355X (round 2, unchanged in round 3): G9 11 / 4096 rays (11 A, 0 B), 48.1 dB; G10 994 / 4096 (961 A, 33 B), 45.3 dB;
    # G16 (round 3; K=192 like G10, but the wide DTU sigmas and range): 70 / 2304 (3.0 %: 70 A, 0 B), 33.8 dB on all rays (115 dB on
    # the rays with the reference's sample set; the random-init field of the fixtures is rough, a differing ray is up to 0.39 off)
    "g9_render_K128": (dict(), 14, 2, 47.0),
    "g10_render_cfg5": (dict(scale=1.75, znear=1.0, zfar=2.5, std_law="facescape"), 1010, 40, 44.0),
    "g16_render_K192_dtu": (dict(), 76, 4, 32.0),
}


def _render_fixture(name):
    g = load(name + ".npz")
    kw = RENDER_FIXTURES[name][0]
    W, H, K, G, n_cand = (int(g[k]) for k in ("W", "H", "K", "G", "n_cand"))
    sc, scene, w, msd, _ = oracle_setup(W, H, int(g["seed"]), **kw)
    NR = g["rays"].shape[0]
    gen = torch.Generator().manual_seed(int(g["noise_seed"]))
    nc, ng, nf = torch.rand(NR, n_cand, generator=gen), torch.randn(NR, G, generator=gen), torch.rand(NR, K, generator=gen)
    assert sha(nc[:64], ng[:64], nf[:64]) == str(g["in_sha"]), "seeded noise not reproducible on this host"
    return g, sc, scene, w, msd, (K, G, n_cand, bool(int(g["white_bkgd"]))), (nc, ng, nf)


@pytest.mark.parametrize("name", sorted(RENDER_FIXTURES))
def test_render_at_metric_sample_counts(ops, precision, name):
    """renderer.forward against the reference's output at the sample counts the metric uses: G9 = K=128 / G=48 on 4096
    rays of the 400x300 bench scene (BASELINE configs[1..3]), G10 = K=192 / G=72, white background, Facescape depth range
    and sigma law (configs[4]), G16 = K=192 / G=72 with the DTU sigma law.  EVERY ray carries a 1e-4 statement:
      (1) with the reference's sample positions every ray matches the reference's colours and depth to 1e-4 (field kernels +
          compositor; K=128 and 192 take the two- / three-samples-per-lane paths of the compositor);
      (2) the HIP sampler reproduces the reference's sample positions to fp32 round-off (3e-6) on every ray except two
          classes that no second implementation of erf can reproduce (the reference's likelihoods come from MKL's vsErf on
          the pinning host, CUDA's erff on an A100, ocml's here: an ulp apart here and there), each verified per ray:
            A. the pick sets differ only by candidates whose likelihood lies within SAT_L of the ray's cut-off
               (helpers.selection_diff; anything further away must match); one such candidate changes the number of
               empty slots and with it every stratified fill sample of the ray;
            B. identical picks, but the gaussian fit (weighted_mean_n_std of the occupancy O) rests on likelihood mass
               that is itself erf round-off residue (sum(O) < 1e-2: a surface just beyond the far plane);
          and for EVERY ray of those classes, not a sample of them:
            a. the G gaussian slots equal the reference implementation's to 3e-6 whenever the fit is conditioned
               (sum(O) >= 1e-2): they depend on O over all candidates, not on the picks (nerf_renderer.py:181-190);
            b. the reference's fill (nerf_renderer.py:377-396, oracle) applied to the HIP pick set with the same noise gives the
               HIP sample positions BIT FOR BIT: what differs from the reference is the pick set alone;
            c. the reference's field + compositor (oracle) evaluated AT THE HIP SAMPLES agree with the HIP colours and depth to 1e-4;
          on the rays with the reference's samples the image matches the reference's to 1e-4, except where a 1e-7 shift of a gaussian
          sample is amplified by the depth positional encoding (200 rad per unit depth) beyond that -- those rays (a handful)
          get statement c as well;
      (3) over all rays: the number of class-A and class-B rays and the PSNR of the image against the reference's image are
          bounded near their measured values (RENDER_FIXTURES).  A class-A/B ray is rendered from a different but equally
          valid random sample set -- like another noise seed -- so PSNR against ground truth is unchanged in expectation."""
    g, sc, scene, w, msd, (K, G, n_cand, white), (nc, ng, nf) = _render_fixture(name)
    _, max_a, max_b, psnr_floor = RENDER_FIXTURES[name]
    hs, hm = hip_scene(ops, sc), hip_mlp(ops, msd)
    rays = T(g["rays"])
    rc = rays.cuda()
    NR = rays.shape[0]
    ref_rgb, ref_d, ref_z = T(g["rgb"]), T(g["depth"]), T(g["z"])

    def errs(rgb, depth, ref_rgb=ref_rgb, ref_d=ref_d):
        return ((rgb.cpu() - ref_rgb).abs().max(-1).values / ref_rgb.abs().max(),
                (depth.cpu() - ref_d).abs() / ref_d.abs().max())

    # (1)
    wts, rgb, depth = ops.render(hs, hm, rc, ref_z.cuda(), white, want_weights=True)
    e_rgb, e_d = errs(rgb, depth)
    print(f"{name} [{precision}] reference z: rgb {e_rgb.max().item():.2e} depth {e_d.max().item():.2e} (all {NR} rays)")
    assert e_rgb.max().item() < TOL and e_d.max().item() < TOL
    np.testing.assert_allclose(wts.cpu().sum(-1).numpy(), g["weights_sum"], atol=3e-5)
    assert max_norm_rel(wts.cpu()[::16], g["weights_sub"]) < TOL
    # (2)
    z, zu = ops.sample_depthguided(hs, rc, K, n_cand, G, 0.05, noise=(nc.cuda(), ng.cuda(), nf.cuda()), want_unfilled=True)
    zh, zuh = z.cpu(), zu.cpu()
    same = torch.isclose(zh, ref_z, rtol=3e-6, atol=1e-7).all(-1)
    zc = O.sample_coarse(rays, n_cand, nc)
    L, Occ = O.point_likelihood(scene, rays, zc)
    ties = set(int(r) for r in g["tie_rays"])      # exact likelihood ties at the cut-off: the reference's pick there is
    n_a = n_b = 0                                  # torch's unstable argsort order (make_golden_r2.py)
    diff = (~same).nonzero().flatten()
    for r in diff.tolist():
        only = set(ref_z[r].tolist()) ^ set(zh[r].tolist())
        cand = [int((zc[r] == zz).nonzero().flatten()[0]) for zz in only if (zc[r] == zz).any()]
        if cand:                                   # class A
            n_a += 1
            if r in ties:
                continue
            Ls = L[r].sort(descending=True).values
            cut = float(Ls[K - G - 1])
            worst = max(abs(float(L[r, c]) - cut) for c in cand)
            assert worst < SAT_L, f"ray {r}: pick differs on a candidate {worst:.1e} away from the cut-off likelihood"
        else:                                      # class B
            n_b += 1
            assert 0 < float(Occ[r].sum()) < 1e-2, f"ray {r}: same picks, well-conditioned gaussian fit, different samples"
    wts, rgb, depth = ops.render(hs, hm, rc, z, white, want_weights=False)
    e_rgb, e_d = errs(rgb, depth)
    hot = (same & ((e_rgb >= TOL) | (e_d >= TOL))).nonzero().flatten()
    assert len(hot) <= 0.002 * NR
    # per-ray statements a / b / c on every ray whose sample set differs (and c on the "hot" ones)
    n_gauss_checked, g_worst = 0, 0.0
    if len(diff):
        rd = rays[diff].contiguous()
        mu, sd = O.weighted_mean_n_std(zc[diff], Occ[diff])
        g_ref = ng[diff] * sd + mu                                                       # nerf_renderer.py:188
        cond = Occ[diff].sum(-1) >= 1e-2
        g_hip = zuh[diff][:, K - G:]
        if cond.any():
            close = torch.isclose(g_hip[cond], g_ref[cond], rtol=3e-6, atol=1e-7)
            g_worst = float(((g_hip[cond] - g_ref[cond]).abs() / g_ref[cond].abs().clamp(min=1e-3)).max())
            n_gauss_checked = int(cond.sum())
            assert close.all(), f"a. gaussian slots of {int((~close.all(-1)).sum())} differing rays are not the reference's (worst {g_worst:.1e})"
        refill = O.fill_up_uniform_samples(zuh[diff], rd, nf[diff])
        assert torch.equal(refill, zh[diff]), "b. the reference's fill of the HIP pick set is not the HIP sample set"
    chk = torch.cat((diff, hot))
    if len(chk):
        o_w, o_rgb, o_d, _ = O.composite(scene, w, rays[chk].contiguous(), zh[chk].contiguous(), white)
        h_rgb, h_d = errs(rgb[chk], depth[chk], o_rgb, o_d)
        print(f"{name} [{precision}] oracle field + compositor AT THE HIP SAMPLES on the {len(diff)} rays with a different sample set and the "
              f"{len(hot)} rays with the reference's samples (to 3e-6) but colours off by more than 1e-4: rgb {h_rgb.max().item():.1e} depth "
              f"{h_d.max().item():.1e}; gaussian slots of {n_gauss_checked} differing rays within {g_worst:.1e} of the reference's; fill of the "
              f"HIP pick sets reproduced bit for bit")
        assert h_rgb.max().item() < TOL and h_d.max().item() < TOL, "c. oracle at the HIP samples"
    cool = same.clone()
    cool[hot] = False
    assert e_rgb[cool].max().item() < TOL and e_d[cool].max().item() < TOL
    # (3)
    mse = (rgb.cpu() - ref_rgb).square().mean().item()
    psnr = 10 * np.log10(1.0 / max(mse, 1e-30))
    mse_same = (rgb.cpu() - ref_rgb)[same].square().mean().item()
    n_diff = int((~same).sum())
    print(f"{name} [{precision}] HIP sampler: {n_diff}/{NR} rays ({100.0 * n_diff / NR:.2f} %) with a different sample set "
          f"(class A {n_a}, class B {n_b}); rays with the reference's samples: rgb {e_rgb[cool].max().item():.2e} depth "
          f"{e_d[cool].max().item():.2e}, PSNR {10 * np.log10(1.0 / max(mse_same, 1e-30)):.1f} dB; ALL rays: rgb "
          f"{e_rgb.max().item():.2e} depth {e_d.max().item():.2e}, PSNR of the image against the reference's {psnr:.1f} dB")
    assert n_a <= max_a and n_b <= max_b, (n_a, n_b)
    assert psnr >= psnr_floor
    # (4) the denominator (G17, oracle/make_golden_seeds.py): the REFERENCE rendered the same rays with two other noise seeds.  A ray
    # whose pick set differs is rendered from other stratified samples -- what another seed does to every ray -- so (i) the HIP image must
    # be at least as close to the reference's image as the reference's own seed-to-seed distance (minus 1 dB), on all rays and on the
    # differing rays alone, and (ii) the mean colour difference over the differing rays must vanish within the seed-to-seed standard
    # error of that mean (no bias: PSNR against ground truth is unchanged in expectation).
    s2s = load("g17_seed_to_seed.npz")
    key = name.split("_")[0]
    seeds = [T(s2s[f"{key}_rgb_s{i}"]) for i in (1, 2)]
    psnr_of = lambda a, b: float(-10.0 * torch.log10((a - b).square().mean().clamp(min=1e-30)))
    s2s_all = max(psnr_of(sr, ref_rgb) for sr in seeds)
    assert abs(s2s_all - max(float(s2s[f"{key}_psnr_s1_vs_fixture"]), float(s2s[f"{key}_psnr_s2_vs_fixture"]))) < 1e-3
    hr = rgb.cpu()
    msg = f"{name} [{precision}] seed-to-seed (reference, two other noise seeds): all rays {s2s_all:.1f} dB vs HIP {psnr:.1f} dB"
    assert psnr >= s2s_all - 1.0
    if len(diff) >= 8:
        d_hip = (hr[diff] - ref_rgb[diff])                                  # (n, 3)
        d_seed = [(sr[diff] - ref_rgb[diff]) for sr in seeds]
        s2s_diff = float(np.mean([psnr_of(sr[diff], ref_rgb[diff]) for sr in seeds]))
        hip_diff = psnr_of(hr[diff], ref_rgb[diff])
        # standard error of the per-ray mean difference under a change of seed, from both seeds' per-ray differences
        per_ray = torch.cat([d.mean(-1) for d in d_seed])
        se = float(per_ray.std() / np.sqrt(len(diff)))
        bias = float(d_hip.mean())
        worst_ray_hip = float(d_hip.abs().max())
        worst_ray_seed = max(float((sr - ref_rgb).abs().max()) for sr in seeds)          # over all rays of the fixture
        msg += (f"; on the {len(diff)} differing rays {s2s_diff:.1f} dB vs HIP {hip_diff:.1f} dB, mean colour difference {bias:+.2e} against a "
                f"seed-to-seed standard error of {se:.2e}, largest per-ray difference {worst_ray_hip:.3f} vs {worst_ray_seed:.3f} between seeds (any ray)")
        if len(diff) >= 64:                    # (a PSNR over a dozen rays is one or two outliers: G9 has 11 differing rays)
            assert hip_diff >= s2s_diff - 1.0
        assert abs(bias) <= 3.0 * se + 1e-6
        assert worst_ray_hip <= worst_ray_seed * 1.05 + 1e-6
    print(msg)
    # (5) PSNR against "ground truth" (G18): the mean of EIGHT other-seed renders of the reference itself (the two of G17 + six more) is the
    # expected image of this scene -- what a ground-truth photo is to the metric of north_star ("PSNR within 0.05 dB of reference on DTU val").
    # PSNR(HIP image, ground truth) must equal PSNR(the reference's fixture image, ground truth) within that margin: the HIP image IS the
    # fixture's image except on the rays whose sample set differs, and those are other draws from the same distribution.
    ens = load("g18_seed_ensemble.npz")
    members = [T(m) for m in ens[f"{key}_rgb"]] + seeds
    if f"{key}_rgb_more" in ens.files:          # round 5: G16's ensemble grown to 16 members (make_golden_seeds.py more g16): a steadier mean
        members += [T(m) for m in ens[f"{key}_rgb_more"]]
    gt = torch.stack(members).mean(0)
    p_hip, p_fix = psnr_of(hr, gt), psnr_of(ref_rgb, gt)
    loo = []
    for i in range(len(members)):                 # the reference's own spread: each member against the mean of the others
        rest = torch.stack([m for j, m in enumerate(members) if j != i]).mean(0)
        loo.append(psnr_of(members[i], rest))
    print(f"{name} [{precision}] PSNR against the {len(members)}-seed ensemble mean of the reference: HIP {p_hip:.3f} dB, reference's fixture image {p_fix:.3f} dB "
          f"(difference {p_hip - p_fix:+.3f} dB; the reference's members against the mean of the others: {min(loo):.2f} .. {max(loo):.2f} dB)")
    assert abs(p_hip - p_fix) <= PSNR_VS_GT_MARGIN_DB[key], (p_hip, p_fix)


def test_cfg5_fp16_mlp_psnr(ops):
    """BASELINE configs[4] asks for an "fp16 MLP on MFMA" at K=192 with a white background: DINER_PRECISION_F16 on the G10
    fixture.  Not inside the 1e-4 parity bar (stated, not hidden): the test states the PSNR of the fp16 image against the
    reference's fp32 image.  north_star allows 0.05 dB on PSNR-vs-ground-truth; an image >= 50 dB from the reference's moves
    a 30 dB PSNR-vs-GT by < 0.05 dB (|dPSNR| <= 20 log10(1 + 10^((30-50)/20)) = 0.83 dB worst case for fully correlated
    errors, ~0.04 dB for uncorrelated ones)."""
    g, sc, scene, w, msd, (K, G, n_cand, white), (nc, ng, nf) = _render_fixture("g10_render_cfg5")
    hs, hm = hip_scene(ops, sc), hip_mlp(ops, msd)
    rc = T(g["rays"]).cuda()
    ref_rgb, ref_d = T(g["rgb"]), T(g["depth"])
    out = {}
    for mode in ("f16", "f16x3"):
        _, rgb, depth = ops.render(hs, hm, rc, T(g["z"]).cuda(), white, precision=mode)
        mse = (rgb.cpu() - ref_rgb).square().mean().item()
        out[mode] = (10 * np.log10(1.0 / max(mse, 1e-30)), max_norm_rel(rgb.cpu(), ref_rgb), max_norm_rel(depth.cpu(), ref_d))
        print(f"cfg5 K=192 white [{mode}]: PSNR vs the reference image {out[mode][0]:.1f} dB, rgb max-norm-rel "
              f"{out[mode][1]:.2e}, depth {out[mode][2]:.2e}")
    assert out["f16"][0] >= 50.0 and out["f16"][1] < 1e-2
    assert out["f16x3"][0] >= 90.0 and out["f16x3"][1] < TOL
    # the metric itself (G18, see test_render_at_metric_sample_counts statement 5): PSNR of the fp16-mode image against the mean of the
    # reference's 8-seed ensemble is the reference image's own to far less than north_star's 0.05 dB -- an image 84 dB from the reference's
    # cannot move a 37 dB PSNR
    ens, s2s = load("g18_seed_ensemble.npz"), load("g17_seed_to_seed.npz")
    gt = torch.stack([T(m) for m in ens["g10_rgb"]] + [T(s2s[f"g10_rgb_s{i}"]) for i in (1, 2)]).mean(0)
    psnr_of = lambda a, b: float(-10.0 * torch.log10((a - b).square().mean()))
    _, rgb16, _ = ops.render(hs, hm, rc, T(g["z"]).cuda(), white, precision="f16")
    d = psnr_of(rgb16.cpu(), gt) - psnr_of(ref_rgb, gt)
    print(f"cfg5 K=192 white [f16]: PSNR against the reference's 8-seed ensemble mean {psnr_of(rgb16.cpu(), gt):.4f} dB, the reference image's own "
          f"{psnr_of(ref_rgb, gt):.4f} dB (difference {d:+.5f} dB)")
    assert abs(d) < 0.005


def test_fp16_overflow_falls_back_to_exact_kernels(ops):
    """Activations beyond the fp16 range (a checkpoint with large hidden values): the fp16-operand kernels raise a device
    flag and the gated exact-fp32 pass recomputes the launch -- no inf / NaN, no host synchronisation; the result is the
    fp32 mode's, bit for bit.  Weights beyond the range the x16 fp16 split supports select the exact kernels up front."""
    g = load("g6_pixelnerf.npz")
    sc, scene, w, msd, rays = oracle_setup(int(g["W"]), int(g["H"]), int(g["seed"]))
    hs = hip_scene(ops, sc)
    pts, dirs = T(g["pts"]).cuda(), T(g["dirs"]).cuda()
    big = {k: v.clone() for k, v in msd.items()}
    big["lin_in.bias"] = big["lin_in.bias"] + 3.0e5 * (torch.arange(512) % 7 == 0)       # hidden activations ~3e5 > 65504
    hm = hip_mlp(ops, big)
    assert hm.h3_ok                                   # the weights themselves are in range (|w| < 1024): bias 3e5 is not
    hm2 = hip_mlp(ops, {k: (v * 1.0) for k, v in msd.items()})
    exact = ops.field_from_points(hs, hm, pts, dirs, precision="fp32")
    for mode in ("f16x3", "f16"):
        got = ops.field_from_points(hs, hm, pts, dirs, precision=mode)
        assert torch.isfinite(got).all()
        assert torch.equal(got, exact), mode
    # in-range network: the flag stays down and the fp16-operand result is NOT the fp32 one bit for bit (different arithmetic)
    a, b = ops.field_from_points(hs, hm2, pts, dirs, precision="f16x3"), ops.field_from_points(hs, hm2, pts, dirs, precision="fp32")
    assert max_norm_rel(a.cpu(), b.cpu()) < TOL_STAGE and not torch.equal(a, b)
    # the fall-back is visible: the handle counts the launches the exact kernels had to recompute
    assert hm.fallback_launches() == 2 and hm2.fallback_launches() == 0
    assert hm.fallback_launches(reset=True) == 2 and hm.fallback_launches() == 0
    # a HIDDEN activation beyond the range (the residual stream stays small): in the per-view blocks, in the post blocks.  The operand
    # that overflows turns its column's products into NaN / inf of either sign; the range check reads the bits of the residual stream in
    # front of lin_out, whatever the signs (until round 3 only NaNs with a clear sign bit got through relu to the check)
    for key, mag in (("blocks.1.fc_0.bias", 1.0e5), ("blocks.3.fc_0.bias", 1.0e5), ("blocks.4.fc_0.bias", 3.0e5)):
        hid = {k: v.clone() for k, v in msd.items()}
        hid[key] = hid[key] + mag * (torch.arange(512) % 7 == 0)
        hh = hip_mlp(ops, hid)
        exact = ops.field_from_points(hs, hh, pts, dirs, precision="fp32")
        for mode in ("f16x3", "f16"):
            assert torch.equal(ops.field_from_points(hs, hh, pts, dirs, precision=mode), exact), (key, mode)
        assert hh.fallback_launches() == 2, key
    # ragged launches (P % 64 != 0: the last post tile holds column groups without points; P % 16 != 0: a ragged 16-point group) with the
    # overflow in the LAST point only -- the flag must come up whichever wave sees it
    n_all = pts.shape[0]
    for P1 in (n_all - (n_all % 64) - 23, 16 * 5 + 3, 7):
        hot = {k: v.clone() for k, v in msd.items()}
        hot["blocks.3.fc_0.bias"] = hot["blocks.3.fc_0.bias"] + 1.0e5 * (torch.arange(512) % 7 == 0)
        hr = hip_mlp(ops, hot)
        exact = ops.field_from_points(hs, hr, pts[:P1], dirs[:P1], precision="fp32")
        for mode in ("f16x3", "f16"):
            assert torch.equal(ops.field_from_points(hs, hr, pts[:P1], dirs[:P1], precision=mode), exact), (P1, mode)
        assert hr.fallback_launches() == 2, P1


def test_scene_prepared_with_another_handle_is_refused(ops):
    """The projected maps carry the lin_z / fc_1 biases of the handle that prepared them (DinerScene.proj_stamp): the C ABI refuses
    them with any other handle instead of silently dropping or duplicating biases; the retired precision value 2 is refused too."""
    import ctypes as C
    g = load("g6_pixelnerf.npz")
    sc, scene, w, msd, rays = oracle_setup(int(g["W"]), int(g["H"]), int(g["seed"]))
    hs = hip_scene(ops, sc)
    hm_a, hm_b = hip_mlp(ops, msd), hip_mlp(ops, msd)
    pts, dirs = T(g["pts"]).cuda(), T(g["dirs"]).cuda()
    out = torch.empty(pts.shape[0], 4, device="cuda")
    ws = torch.empty(ops.lib.diner_field_workspace_bytes(pts.shape[0]), dtype=torch.uint8, device="cuda")
    hs.prepare(hm_a)

    def call(handle, prec):
        return ops.lib.diner_field_from_points_f32(hs.ref, handle, C.c_void_p(pts.data_ptr()), C.c_void_p(dirs.data_ptr()),
                                                   pts.shape[0], prec, C.c_void_p(out.data_ptr()), C.c_void_p(ws.data_ptr()),
                                                   C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert call(hm_a.handle, ops.PRECISION_F16X3) == 0
    assert call(hm_b.handle, ops.PRECISION_F16X3) == -1 and b"another packed-weights handle" in ops.lib.diner_last_error()
    assert call(hm_a.handle, 2) == -1 and b"retired" in ops.lib.diner_last_error()
    # ABI v4: the plain-fp16 mode gathers from the fp16 copy of the projected maps; without it the call is refused (nothing was prepared
    # for that mode yet: this scene has only been used with f16x3), and the copy is stale as soon as the maps are prepared again
    assert hs.struct.latent_proj_f16 is None
    assert call(hm_a.handle, ops.PRECISION_F16) == -1 and b"diner_scene_prepare_f16" in ops.lib.diner_last_error()
    hs.prepare(hm_a, f16=True)
    assert hs.struct.latent_proj_f16 and call(hm_a.handle, ops.PRECISION_F16) == 0
    first = out.clone()
    hs.prepare(hm_a, force=True)                     # new fp32 maps: the host marks the fp16 copy stale and rebuilds it on the next f16 call
    assert not hs._f16_current
    assert torch.equal(ops.field_from_points(hs, hm_a, pts, dirs, precision="f16"), first) and hs._f16_current
    with pytest.raises(ValueError):
        ops.set_precision(2)
    assert max_norm_rel(ops.field_from_points(hs, hm_b, pts, dirs).cpu(), g["out"]) < TOL_STAGE      # the host re-prepares


def test_freq_factor_is_honoured(ops, precision):
    """PositionalEncoding.freq_factor travels with the packed-weights handle into the field kernels (it used to be a
    hard-coded 6.28): a network evaluated with freq_factor = 3.14 matches the oracle run with 3.14."""
    g = load("g6_pixelnerf.npz")
    sc, scene, w, msd, rays = oracle_setup(int(g["W"]), int(g["H"]), int(g["seed"]))
    hs = hip_scene(ops, sc)
    hm = ops.HipMlp({k: v.cuda() for k, v in msd.items()}, freq_factor=3.14)
    pts, dirs = T(g["pts"]), T(g["dirs"])
    want = O.mlp_forward(w, O.mlp_input(scene, pts, dirs, 6, 3.14))
    want = torch.cat((torch.sigmoid(want[:, :3]), torch.relu(want[:, 3:])), -1)
    got = ops.field_from_points(hs, hm, pts.cuda(), dirs.cuda()).cpu()
    assert max_norm_rel(got, want) < TOL_STAGE
    assert max_norm_rel(got, g["out"]) > 1e-2          # and it is not the 6.28 result


def test_helpers_against_reference_fixture(ops):
    """Rows f2 / f3 against outputs of the reference's OWN functions (oracle/make_golden_r2.py, G11): torch_cmap
    (torch_helpers.py:42-75), depth2normal on maps with holes (depth2normal.py:7-87), gen_rays (cam_geometry.py:5-48)."""
    from diner_amd import imageio
    g = load("g11_helpers.npz")
    # depth colour map: the reference returns float64 colours; its 8-bit image (save_image) is what gets written
    depth = T(g["cmap_depth"])
    for tag, (vmin, vmax) in dict(auto=(None, None), fixed=(0.25, 1.75), hi=(None, 2.0)).items():
        want = (T(g["cmap_" + tag]) * 255 + 0.5).clamp(0, 255).to(torch.uint8)                 # (2,3,H,W)
        for b in range(2):
            # vmin / vmax None: per-image min / max, computed on the device like torch_cmap does per batch element
            got = imageio.depth_to_uint8(depth[b].cuda(), vmin=vmin, vmax=vmax).cpu()          # (H,W,3)
            assert torch.equal(got, want[b].permute(1, 2, 0)), (tag, b)
    # depth2normal
    for dm, Km, want in ((T(g["d2n_depth"]), T(g["d2n_K"]), T(g["d2n_normals"])),):
        got = ops.depth2normal(dm.cuda(), Km.cuda()).cpu()
        assert torch.equal(torch.isnan(got), torch.isnan(want))
        fin = ~torch.isnan(want)
        assert (got[fin] - want[fin]).abs().max().item() <= 2e-6
    sc = oracle_setup(48, 40, int(g["d2n_scene_seed"]))[0]
    got = ops.depth2normal(sc["depths"].cuda(), sc["src_intrinsics"].cuda()).cpu()
    want = T(g["d2n_scene_normals"])
    assert torch.equal(torch.isnan(got), torch.isnan(want))
    assert (got[~torch.isnan(want)] - want[~torch.isnan(want)]).abs().max().item() <= 2e-6
    # gen_rays
    W, H = int(g["rays_W"]), int(g["rays_H"])
    got = ops.gen_rays(T(g["rays_E"]), T(g["rays_K"]), W, H, T(g["rays_near"]), T(g["rays_far"]), "cuda").cpu()
    want = T(g["rays"]).view(3, H * W, 8)
    assert (got - want).abs().max().item() <= 5e-7 and torch.equal(got[..., 6:], want[..., 6:])


def test_ray_batch_split_invariance(ops, precision):
    """diner.py:85 splits rays into batches; results must not depend on the split (bit-exact)."""
    sc, scene, w, msd, rays = oracle_setup(32, 32, 3)
    hs, hm = hip_scene(ops, sc), hip_mlp(ops, msd)
    rc = rays.cuda()
    gen = torch.Generator().manual_seed(5)
    noise = (torch.rand(1024, 1000, generator=gen).cuda(), torch.randn(1024, 24, generator=gen).cuda(),
             torch.rand(1024, 64, generator=gen).cuda())
    z_all = ops.sample_depthguided(hs, rc, 64, 1000, 24, noise=noise)
    _, rgb_all, d_all = ops.render(hs, hm, rc, z_all, True)
    parts = []
    for a, b in ((0, 100), (100, 611), (611, 1024)):
        zp = ops.sample_depthguided(hs, rc[a:b], 64, 1000, 24, noise=tuple(n[a:b] for n in noise))
        assert torch.equal(zp, z_all[a:b])
        parts.append(ops.render(hs, hm, rc[a:b], zp, True)[1])
    assert torch.equal(torch.cat(parts), rgb_all)


def test_philox_sampler_statistics(ops):
    """Production noise (in-kernel Philox): z sorted, inside [near, far] for fill samples, reproducible per seed."""
    sc, scene, w, msd, rays = oracle_setup(32, 32, 3)
    hs = hip_scene(ops, sc)
    rc = rays.cuda()
    z1 = ops.sample_depthguided(hs, rc, 128, 1000, 48, seed=11)
    z2 = ops.sample_depthguided(hs, rc, 128, 1000, 48, seed=11)
    z3 = ops.sample_depthguided(hs, rc, 128, 1000, 48, seed=12)
    assert torch.equal(z1, z2) and not torch.equal(z1, z3)
    assert (z1[:, 1:] >= z1[:, :-1]).all() and torch.isfinite(z1).all()
    # rays that see no surface are a pure stratification of [near, far]
    L, Oo = O.point_likelihood(scene, rays, O.sample_coarse(rays, 1000, torch.full((1024, 1000), 0.5)))
    empty = ~(Oo != 0).any(-1)
    if empty.any():
        ze = z1.cpu()[empty]
        edges = torch.linspace(sc["znear"], sc["zfar"], 129)
        assert (ze >= edges[:-1] - 1e-5).all() and (ze <= edges[1:] + 1e-5).all()


def test_replicated_points_are_bit_identical(ops, precision):
    """Several tiles per workgroup: the 512 fixture points repeated 48 times (1536 tiles over 256 workgroups: six tiles
    per workgroup).  Identical inputs must give bit-identical outputs whatever tile slot, workgroup or pipeline phase
    they land in, and replica 0 must still match the reference.  (tools/stress_repeat.py: the same over 40 runs and 400
    replicas in every mode.)"""
    g = load("g6_pixelnerf.npz")
    sc, scene, w, msd, rays = oracle_setup(int(g["W"]), int(g["H"]), int(g["seed"]))
    hs, hm = hip_scene(ops, sc), hip_mlp(ops, msd)
    R = 48
    pts, dirs = T(g["pts"]).repeat(R, 1).cuda(), T(g["dirs"]).repeat(R, 1).cuda()
    out = ops.field_from_points(hs, hm, pts, dirs).cpu().view(R, -1, 4)
    assert max_norm_rel(out[0], g["out"]) < TOL_STAGE
    for r in range(1, R):
        assert torch.equal(out[r], out[0]), f"replica {r} differs from replica 0 [{precision}]"


def test_depth2normal_against_reference_ops(ops):
    """Row f2: the HIP depth2normal against the torch restatement that make_golden.py checks bit-exact against the
    reference (src/util/depth2normal.py on host tensors): synthetic scene depths (sphere + plane + background), plus a
    map with isolated holes, a hole on the border and a pixel column where the back-projected x is exactly 0."""
    from src.util.depth2normal import depth2normal
    sc = oracle_setup(48, 40, 3)[0]
    d = sc["depths"].clone()                                  # (4,1,H,W)
    K = sc["src_intrinsics"].clone()
    g = torch.Generator().manual_seed(5)
    d2 = torch.rand(2, 1, 33, 47, generator=g) + 0.5
    d2[0, 0, 5, 7] = 0; d2[0, 0, 0, 3] = 0; d2[0, 0, 32, 46] = 0; d2[1, 0, 10:14, 20:23] = 0; d2[1, 0, :, 0] = 0
    K2 = torch.tensor([[[40.0, 0, 23.5], [0, 42.0, 16.5], [0, 0, 1]], [[38.0, 0, 20.0], [0, 38.0, 15.0], [0, 0, 1]]])
    for dm, Km in ((d, K), (d2, K2)):
        want = depth2normal(dm, Km)                          # host tensors: the reference's ops
        got = ops.depth2normal(dm.cuda(), Km.cuda()).cpu()
        assert torch.equal(torch.isnan(got), torch.isnan(want))
        fin = ~torch.isnan(want)
        diff = (got[fin] - want[fin]).abs().max().item()
        frac = (got[fin] != want[fin]).float().mean().item()
        print(f"depth2normal {tuple(dm.shape)}: max abs diff {diff:.2e}, {100 * frac:.3f} % of values differ in the last bits")
        assert diff <= 2e-6                                   # unit vectors: <= a few ulp (cross / norm association)
        assert torch.equal(got[:, :, dm[:, 0].eq(0).any(0)].isnan(), want[:, :, dm[:, 0].eq(0).any(0)].isnan())
    assert torch.equal(ops.depth2normal(d.cuda(), K.cuda()), depth2normal(d.cuda(), K.cuda()))   # module dispatch


def test_gen_rays_against_reference_ops(ops):
    """Row f3: device ray generation against the reference's torch ops (cam_geometry.py:5-48), whole images and the
    ray ranges a sharded rank asks for (ragged last shard included)."""
    from src.util.cam_geometry import gen_rays
    from diner_amd.render import shard_range
    sc = oracle_setup(40, 30, 1)[0]
    E = torch.stack([sc["target_extrinsics"].view(4, 4), sc["src_extrinsics"][1], sc["src_extrinsics"][3]])
    Km = torch.stack([sc["target_intrinsics"].view(3, 3), sc["src_intrinsics"][1], sc["src_intrinsics"][3]])
    W, H = 37, 29
    zn, zf = torch.tensor([0.5, 0.6, 0.7]), torch.tensor([1.5, 1.6, 1.7])
    want = gen_rays(E, Km, W, H, zn, zf).view(3, H * W, 8)
    got = ops.gen_rays(E, Km, W, H, zn, zf, "cuda").cpu()
    err = (got - want).abs().max().item()
    print(f"gen_rays {W}x{H}: max abs diff {err:.2e}")
    assert err <= 5e-7
    assert torch.equal(got[..., :3], want[..., :3].expand_as(got[..., :3])) or (got[..., :3] - want[..., :3]).abs().max() <= 2e-7
    assert torch.equal(got[..., 6:], want[..., 6:])
    for world in (3, 8):
        parts = [ops.gen_rays(E, Km, W, H, zn, zf, "cuda", *(lambda lo, hi: (lo, hi - lo))(*shard_range(H * W, r, world)))
                 for r in range(world)]
        assert torch.equal(torch.cat(parts, dim=1).cpu(), got)
    assert ops.gen_rays(E, Km, W, H, zn, zf, "cuda", ray0=H * W, n_rays=0).shape == (3, 0, 8)
    with pytest.raises(RuntimeError):
        ops.gen_rays(E, Km, W, H, zn, zf, "cuda", ray0=H * W - 3, n_rays=4)
    assert torch.equal(gen_rays(E.cuda(), Km.cuda(), W, H, zn.cuda(), zf.cuda()).cpu().view(3, -1, 8), got)   # module dispatch


def test_edge_cases_empty_ragged_and_limits(ops, precision):
    """Empty inputs, ragged point counts (1, 15, 17, 1000 points: partial 16-point tiles and partial 64-point groups),
    rays that see nothing (every likelihood zero: pure stratified fill), and the documented size limits."""
    sc, scene, w, msd, rays = oracle_setup(32, 32, 2)
    hs, hm = hip_scene(ops, sc), hip_mlp(ops, msd)
    rc = rays.cuda()
    # --- empty
    z0 = ops.sample_depthguided(hs, rc[:0], 64, 1000, 24)
    assert z0.shape == (0, 64)
    wts, rgb, dep = ops.render(hs, hm, rc[:0], z0, True)
    assert rgb.shape == (0, 3) and dep.shape == (0,)
    assert ops.field_from_points(hs, hm, torch.zeros(0, 3).cuda(), torch.zeros(0, 3).cuda()).shape == (0, 4)
    # --- ragged point counts: every prefix equals the prefix of the big call, bit for bit
    g = load("g6_pixelnerf.npz")
    sc6, _, _, msd6, _ = oracle_setup(int(g["W"]), int(g["H"]), int(g["seed"]))
    hs6, hm6 = hip_scene(ops, sc6), hip_mlp(ops, msd6)
    pts, dirs = T(g["pts"]).repeat(2, 1).cuda(), T(g["dirs"]).repeat(2, 1).cuda()      # 1024 points
    full = ops.field_from_points(hs6, hm6, pts, dirs)
    assert max_norm_rel(full[:512].cpu(), g["out"]) < TOL_STAGE
    for n in (1, 15, 17, 63, 65, 1000):
        part = ops.field_from_points(hs6, hm6, pts[:n], dirs[:n])
        assert torch.equal(part, full[:n]), f"{n} points [{precision}]"
    # --- rays that miss every surface: all samples come from the stratified fill, ascending inside [near, far]
    away = rc[:64].clone()
    away[:, 3:6] = -away[:, 3:6]                              # look away from the scene
    gen = torch.Generator().manual_seed(9)
    noise = (torch.rand(64, 1000, generator=gen).cuda(), torch.randn(64, 24, generator=gen).cuda(),
             torch.rand(64, 64, generator=gen).cuda())
    z, zu = ops.sample_depthguided(hs, away, 64, 1000, 24, noise=noise, want_unfilled=True)
    assert (zu == 0).all()
    assert (z[:, 1:] >= z[:, :-1]).all() and (z >= away[:, 6:7]).all() and (z <= away[:, 7:8]).all()
    edges = away[:, 6:7] + (away[:, 7:8] - away[:, 6:7]) * torch.arange(65, device="cuda") / 64
    assert ((z >= edges[:, :-1] - 1e-6) & (z <= edges[:, 1:] + 1e-6)).all()            # one sample per stratum
    _, rgb, dep = ops.render(hs, hm, away, z, True)
    assert torch.isfinite(rgb).all() and torch.isfinite(dep).all()
    # --- limits
    ops.sample_depthguided(hs, rc[:8], 256, 1024, 96)                                  # largest supported sizes
    for bad in ((257, 1000, 96), (64, 1025, 24), (64, 1000, 65)):
        with pytest.raises(RuntimeError):
            ops.sample_depthguided(hs, rc[:8], *bad)
    with pytest.raises(ValueError):
        ops.mlp_forward(hm, torch.zeros(3, 8, 567).cuda())                             # NV != 4


def test_plain_fp16_mode_accuracy(ops):
    """DINER_PRECISION_F16: plain fp16 operands, fp32 accumulation (BASELINE configs[4], "fp16 MLP on MFMA").  This mode
    is NOT inside the 1e-4 parity bar and is never the default; the test states what it delivers: max-norm relative
    error below 5e-3 on PixelNeRF.forward and on the end-to-end render with the reference's sample positions."""
    prev = ops.get_precision()
    ops.set_precision(ops.PRECISION_F16)
    try:
        g = load("g6_pixelnerf.npz")
        sc, scene, w, msd, rays = oracle_setup(int(g["W"]), int(g["H"]), int(g["seed"]))
        hs, hm = hip_scene(ops, sc), hip_mlp(ops, msd)
        out = ops.field_from_points(hs, hm, T(g["pts"]).cuda(), T(g["dirs"]).cuda()).cpu()
        e_field = max_norm_rel(out, g["out"])
        g8 = load("g8_render_cfg1.npz")
        sc8, _, _, msd8, _ = oracle_setup(int(g8["W"]), int(g8["H"]), int(g8["seed"]))
        hs8, hm8 = hip_scene(ops, sc8), hip_mlp(ops, msd8)
        r8, z8 = T(g8["rays"]).cuda(), T(g8["z"]).cuda()
        _, rgb, dep = ops.render(hs8, hm8, r8, z8, False)
        e_rgb = max_norm_rel(rgb.cpu(), g8["rgb"])
        print(f"plain fp16 operands: PixelNeRF.forward {e_field:.2e}, e2e rgb {e_rgb:.2e} (max-norm relative)")
        assert 1e-5 < e_field < 5e-3 and e_rgb < 5e-3
    finally:
        ops.set_precision(prev)


def test_image_output_against_reference_ops(ops, tmp_path):
    """Row f3: quantisation and depth colour map on the device against the reference's host path restated with numpy /
    matplotlib (torch_cmap torch_helpers.py:42-75 + save_image's uint8(clamp(v*255+0.5, 0, 255))), bit-exact, and the PNG
    files written from them."""
    import matplotlib.pyplot as plt
    from diner_amd import imageio
    g = torch.Generator().manual_seed(12)
    rgb = torch.rand(3, 45, 61, generator=g) * 1.2 - 0.1               # some values outside [0, 1]
    want = (rgb * 255 + 0.5).clamp(0, 255).permute(1, 2, 0).to(torch.uint8)
    got = imageio.to_uint8(rgb.cuda()).cpu()
    assert torch.equal(got, want)
    depth = torch.rand(1, 45, 61, generator=g) * 0.9 + 0.5
    depth[0, 3, 4] = depth.max()                                        # x == 1 exactly -> last table entry
    for vmin, vmax in ((None, None), (0.25, 1.75), (None, 2.0)):
        x = depth.numpy().astype(float)[None]                           # (B,1,H,W) float64, as torch_cmap does
        lo = vmin if vmin else np.min(x.reshape(1, -1), axis=-1).reshape(-1, 1, 1, 1)
        hi = vmax if vmax else np.max(x.reshape(1, -1), axis=-1).reshape(-1, 1, 1, 1)
        col = torch.from_numpy(plt.get_cmap("viridis")(((x - lo) / (hi - lo))[:, 0])[..., :3])[0]      # (H,W,3) float64
        want_d = (col * 255 + 0.5).clamp(0, 255).to(torch.uint8)
        got_d = imageio.depth_to_uint8(depth.cuda(), vmin=vmin, vmax=vmax).cpu()
        assert torch.equal(got_d, want_d), (vmin, vmax)
    imageio.save_prediction(str(tmp_path), "v0", rgb.cuda(), depth.cuda())
    assert np.array_equal(imageio.read_png(str(tmp_path / "v0_pred.png")), want.numpy())
    assert imageio.read_png(str(tmp_path / "v0_depth.png")).shape == (45, 61, 3)


def test_full_frame_properties_800x600(ops):
    """BASELINE configs[2]/[3] at FULL size (one 800x600 frame = 480,000 rays, 128 samples per ray): no reference output exists at this
    size, so the statements are the size-independent properties of the path: (1) the frame does not depend on how its rays are batched or
    sharded -- rendering the contiguous ray ranges of 2, 7 and 8 "ranks" (diner_amd.render.shard_range, ragged last shard) with ragged batch
    sizes and concatenating gives the single-pass frame bit for bit (one noise seed per frame, keyed by ray position); (2) the same seed reproduces the frame bit for bit, another seed does
    not; (3) compositing weights are a sub-partition of unity (0 <= sum(w) <= 1 + eps), colours stay in [0, 1] on a black background,
    expected depth lies inside [near, far]; (4) with a white background every ray gains exactly 1 - sum(w) in all three channels."""
    from diner_amd.render import shard_range
    W, H, K, G, n_cand = 800, 600, 128, 48, 1000
    sc, scene, w, msd, rays = oracle_setup(W, H, 0)
    hs, hm = hip_scene(ops, sc), hip_mlp(ops, msd)
    NR = W * H
    E, Km = sc["target_extrinsics"][None], sc["target_intrinsics"][None]

    def render_range(lo, hi, batch, seed, white=False, want_w=False):
        r = ops.gen_rays(E, Km, W, H, sc["znear"], sc["zfar"], "cuda", ray0=lo, n_rays=hi - lo)[0]
        outs = []
        for r0 in range(0, hi - lo, batch):
            rb = r[r0:r0 + batch]
            # one seed per frame; the in-kernel noise of a ray is keyed by (seed, index of the ray in the frame)
            z = ops.sample_depthguided(hs, rb, K, n_cand, G, 0.05, noise=None, seed=seed, ray_index0=lo + r0)
            wts, rgb, dep = ops.render(hs, hm, rb, z, white, want_weights=want_w)
            outs.append(torch.cat((rgb, dep[:, None]) + ((wts.sum(-1, keepdim=True), z[:, -1:], z[:, :1]) if want_w else ()), dim=-1))
        return torch.cat(outs)

    full = render_range(0, NR, 8192, seed=3, want_w=True)
    assert full.shape == (NR, 7) and torch.isfinite(full).all()
    # (1) sharding / batching invariance: the ray ranges of 2 and 7 ranks (ragged last shard), each with its own ragged batch size
    for world, batch in ((2, 5000), (7, 8192 + 17), (8, 8192)):        # (8 = BASELINE configs[3]'s rank count)
        parts = [render_range(*shard_range(NR, r, world), batch, seed=3) for r in range(world)]
        assert torch.equal(torch.cat(parts), full[:, :4]), f"{world}-way sharded frame differs"
    # (2) determinism
    again = render_range(0, 65536, 8192, seed=3)
    other = render_range(0, 65536, 8192, seed=4)
    assert torch.equal(again, full[:65536, :4]) and not torch.equal(other, full[:65536, :4])
    # (3) ranges.  Gaussian samples are not clamped to [near, far] (nerf_renderer.py:185-189): a ray whose last sample lies beyond `far`
    #     gets a negative last interval, hence a negative alpha and weight, exactly as in the reference -- the
    #     partition-of-unity bound holds for all other rays
    rgb, dep, wsum, zlast, zfirst = full[:, :3], full[:, 3], full[:, 4], full[:, 5], full[:, 6]
    inside = (zlast <= sc["zfar"]) & (zfirst >= sc["znear"])
    assert int(inside.sum()) > 0.5 * NR             # (the back plane of the synthetic scene sits at the far end of the range)
    assert float(wsum[inside].min()) >= -1e-5 and float(wsum[inside].max()) <= 1.0 + 1e-4
    assert float(rgb[inside].min()) >= -1e-5 and float(rgb[inside].max()) <= 1.0 + 1e-4
    # expected depth = sum(w z) with w >= 0 and near <= z <= far:  near * sum(w) <= depth <= far * sum(w)
    assert bool((dep[inside] >= sc["znear"] * wsum[inside] - 1e-4).all()) and bool((dep[inside] <= sc["zfar"] * wsum[inside] + 1e-4).all())
    hit = inside & (wsum > 0.5)
    # (4) white background
    sub = slice(200000, 200000 + 16384)
    white = render_range(200000, 200000 + 16384, 8192, seed=3, white=True)
    black = full[sub]
    assert torch.allclose(white[:, :3], black[:, :3] + (1 - black[:, 4:5]), atol=2e-6)
    assert torch.equal(white[:, 3], black[:, 3])
    print(f"800x600 frame: {int((~inside).sum())} rays with a sample beyond far; on the others sum(w) in [{float(wsum[inside].min()):.2e}, "
          f"{float(wsum[inside].max()):.6f}]; {int(hit.sum())} of {NR} rays have sum(w) > 0.5")


def test_full_frame_properties_cfg5_1024(ops):
    """BASELINE configs[4] at FULL size (1024 x 1024 target, 192 samples per ray = 201 M sample points, Facescape depth range and sigma law,
    white background; reference: python_scripts/create_prediction_folder.py:44-47 with configs/evaluate_diner_on_facescape.yaml), in the
    arithmetic that config names (plain fp16 operands) and in the parity-grade one (f16x3).  No reference output exists at this size: the
    statements are size-independent properties.  (1) Sharding / batching invariance, bit for bit, in BOTH modes: the ray ranges of 8 ranks
    with a ragged batch size against the single-pass frame (K = 192 is the sample count whose 12 depth segments per ray go through the
    segment-aware tile queues, and the plain-fp16 mode gathers from the fp16 copy of the projected maps).  (2) White background: every ray
    gains exactly 1 - sum(w).  (3) The fp16-mode frame against the f16x3 frame: PSNR >= 60 dB (the mode's ~5e-4), same sample positions."""
    from diner_amd.render import shard_range
    W = H = 1024
    K, G, n_cand = 192, 72, 1000
    sc, scene, w, msd, rays = oracle_setup(W, H, 0, scale=1.75, znear=1.0, zfar=2.5, std_law="facescape")
    hs, hm = hip_scene(ops, sc), hip_mlp(ops, msd)
    NR = W * H
    E, Km = sc["target_extrinsics"][None], sc["target_intrinsics"][None]

    def render_range(lo, hi, batch, seed, mode, white=True):
        r = ops.gen_rays(E, Km, W, H, sc["znear"], sc["zfar"], "cuda", ray0=lo, n_rays=hi - lo)[0]
        outs = []
        for r0 in range(0, hi - lo, batch):
            rb = r[r0:r0 + batch]
            z = ops.sample_depthguided(hs, rb, K, n_cand, G, 0.05, noise=None, seed=seed, ray_index0=lo + r0)
            wts, rgb, dep = ops.render(hs, hm, rb, z, white, want_weights=True, precision=mode)
            outs.append(torch.cat((rgb, dep[:, None], wts.sum(-1, keepdim=True)), dim=-1))
        return torch.cat(outs)

    frames = {}
    for mode in ("f16", "f16x3"):
        full = render_range(0, NR, 5461, seed=11, mode=mode)              # (5461 rays x 192 = one launch of 2^20 points, ragged last batch)
        assert full.shape == (NR, 5) and torch.isfinite(full).all(), mode
        parts = [render_range(*shard_range(NR, r, 8), 4096 + 13, seed=11, mode=mode) for r in (0, 3, 7)]
        for r, part in zip((0, 3, 7), parts):
            lo, hi = shard_range(NR, r, 8)
            assert torch.equal(part, full[lo:hi]), f"{mode}: shard {r} of 8 differs from the single-pass frame"
        sub = slice(500000, 500000 + 8192)
        black = render_range(500000, 500000 + 8192, 8192, seed=11, mode=mode, white=False)
        assert torch.allclose(full[sub, :3], black[:, :3] + (1 - black[:, 4:5]), atol=4e-6 if mode == "f16x3" else 1e-5), mode
        assert torch.equal(full[sub, 3], black[:, 3])
        frames[mode] = full
    assert hm.fallback_launches() == 0
    mse = (frames["f16"][:, :3] - frames["f16x3"][:, :3]).square().mean().item()
    psnr = 10 * np.log10(1.0 / max(mse, 1e-30))
    print(f"1024x1024 K=192 frame: plain fp16 against f16x3 {psnr:.1f} dB, largest colour difference "
          f"{float((frames['f16'][:, :3] - frames['f16x3'][:, :3]).abs().max()):.2e}")
    assert psnr >= 60.0


def test_tile_queues_hand_out_every_tile_once(ops):
    """Round 4: the per-view kernel's tiles are dealt to the 8 XCD queues by depth segment for ANY samples-per-ray count that is a multiple of
    16 (QueueMap, mlp_h3n.hip: ray groups of 8 / gcd(S, 8), runs of S R / 8 slots per queue, entries past the launch skipped, the workgroups'
    first tiles on the same map, stealing from the other queues).  A tile that is skipped leaves its 16 points unwritten, one that is handed out
    twice is harmless -- so every (rays, samples) shape below is rendered by the f16x3 kernels (dynamic queues) and by the exact-fp32 kernels
    (static round-robin over tiles) from a NaN-filled output: all finite and equal to the fp32 mode within the arithmetic's 2e-5.  Shapes: fewer
    tiles than workgroups, ragged last ray group, S odd / even / a multiple of 8 / not a multiple of 16 (the default map), K up to the limit."""
    sc, scene, w, msd, rays = oracle_setup(48, 40, 5)
    hs, hm = hip_scene(ops, sc), hip_mlp(ops, msd)
    g = torch.Generator().manual_seed(9)
    for NR, K in ((1, 16), (3, 48), (7, 80), (333, 192), (1000, 16), (257, 112), (96, 256), (41, 40), (130, 128), (64, 208)):
        r = rays[torch.randint(0, rays.shape[0], (NR,), generator=g)].cuda()
        z = (r[:, 6:7].cpu() + (r[:, 7:8] - r[:, 6:7]).cpu() * torch.rand(NR, K, generator=g).sort(-1).values).cuda()
        exact = ops.field_from_rays(hs, hm, r, z, precision="fp32")
        ws_bytes = int(ops.lib.diner_field_workspace_bytes(NR * K))
        for mode in ("f16x3", "f16"):
            # poison what the call is about to allocate (the caching allocator hands the freed blocks of exactly these sizes back): the
            # output and the hand-over workspace of the previous call would otherwise still hold a correct result
            junk = [torch.full((NR, K, 4), float("nan"), device="cuda"), torch.full((ws_bytes,), 0xFF, dtype=torch.uint8, device="cuda")]
            torch.cuda.synchronize()
            del junk
            got = ops.field_from_rays(hs, hm, r, z, precision=mode)
            assert torch.isfinite(got).all(), (NR, K, mode)
            tol = 2e-5 if mode == "f16x3" else 2e-2
            assert max_norm_rel(got.cpu(), exact.cpu()) < tol, (NR, K, mode)
    assert hm.fallback_launches() == 0


def _g20_inputs(variant):
    """The realistic-magnitude scene of G20 rebuilt from its seeds (sha256-guarded): (fixture, scene dict with latent, mlp state dict)."""
    from diner_amd.synthetic import make_scene, realistic_latent, realistic_mlp_state_dict
    from src.util.depth2normal import depth2normal
    g = load("g20_realistic.npz")
    W, H = int(g["W"]), int(g["H"])
    sc = make_scene(W, H, seed=int(g["scene_seed"]), latent=False)
    sc["normals"] = depth2normal(sc["depths"], sc["src_intrinsics"])
    Hf = (H + 128) // 2
    lat = realistic_latent(4, 512, Hf, Hf, int(g["latent_seed"]), hot_gain=float(g[f"hot_gain_{variant}"]))
    assert sha(lat[:, :8, :4, :4], lat[:, -8:, -4:, -4:]) == str(g[f"lat_sha_{variant}"]), "realistic_latent is not the fixture's on this host"
    sc["latent"] = lat
    msd = realistic_mlp_state_dict(int(g["mlp_seed"]))
    assert sha(*[msd[k] for k in sorted(msd)]) == str(g["mlp_sha"]), "realistic_mlp_state_dict is not the fixture's on this host"
    return g, sc, msd


@pytest.mark.parametrize("variant", ["a", "b"], ids=["realistic_magnitudes", "beyond_the_fp16_range_inside_a_render"])
def test_realistic_magnitudes_g20(ops, precision, variant):
    """G20 (round 6): the reference's renderer.forward on a latent with ResNet-feature statistics (non-negative, heavy-tailed, a few dominant
    channels: mean 0.74, maximum 470) and an MLP that is not the init (row-wise scales, |w| up to 50, biases of O(1)) -- K = 128, 512 rays,
    the reference's sample positions, both parity-grade arithmetic modes at 1e-4 (oracle/make_golden_realistic.py; resnetfc.py:129-159,
    image_encoder.py:97-146).  Variant a: residual stream up to 5e3, hidden activations up to 1.2e3 -- the f16x3 kernels must NOT fall back
    (fallback_launches() == 0: the headline mode is the mode such a checkpoint runs in).  Variant b: the hot channels x 128, hidden
    activations up to 1.1e5 (fp16 ends at 65504) -- the fp16-operand kernels must raise their flag INSIDE a full render and the gated
    exact-fp32 pass must deliver the reference's values (until round 6 only hand-planted values exercised it)."""
    g, sc, msd = _g20_inputs(variant)
    hs, hm = hip_scene(ops, sc), hip_mlp(ops, msd)
    assert hm.h3_ok and hm.wmax < 64, hm.wmax
    rays, z = T(g["rays"]).cuda(), T(g["z"]).cuda()
    K = int(g["K"])
    hm.fallback_launches(reset=True)
    field = ops.field_from_rays(hs, hm, rays, z).cpu()                       # (NR, K, 4)
    wts, rgb, depth = ops.render(hs, hm, rays, z, False, want_weights=True)
    fb = hm.fallback_launches(reset=True)
    ref_f = T(g[f"field_{variant}"]).view(-1, K, 4)
    e_sig = ((field[::4, :, 3] - ref_f[..., 3]).abs().max() / ref_f[..., 3].abs().max()).item()
    e_col = (field[::4, :, :3] - ref_f[..., :3]).abs().max().item()
    ref_rgb, ref_d = T(g[f"rgb_{variant}"]), T(g[f"depth_{variant}"])
    ok = torch.isfinite(ref_rgb).all(-1) & torch.isfinite(ref_d)            # (variant b: the REFERENCE's compositor overflows fp32 on a few rays
    n_bad = int((~ok).sum())                                                 # whose last sample lies beyond `far`: exp(+|delta| sigma) at sigma ~ 5e3)
    e_rgb = ((rgb.cpu() - ref_rgb)[ok].abs().max() / ref_rgb[ok].abs().max()).item()
    e_d = ((depth.cpu() - ref_d)[ok].abs().max() / ref_d[ok].abs().max()).item()
    print(f"G20 {variant} [{precision}]: field sigma {e_sig:.2e} (max-norm-rel, sigma up to {float(ref_f[..., 3].max()):.3g}), colours {e_col:.2e} (abs), "
          f"rgb {e_rgb:.2e}, depth {e_d:.2e} on {int(ok.sum())} rays ({n_bad} rays non-finite in the reference itself); fall-back launches {fb}; "
          f"reference activations: residual stream up to {float(g[f'stream_max_{variant}']):.3g}, hidden up to {float(g[f'hidden_max_{variant}']):.3g}")
    # The bar: 1e-4 -- except where fp32 itself cannot deliver it: the fixture carries the distance between the REFERENCE's fp32 field and a
    # float64 evaluation of the same network on the same inputs (variant a: colours 7.7e-5, variant b: 1.3e-3 -- a residual stream of 6e5 in
    # front of a sigmoid); a second fp32 implementation (another summation order) is held to 3 x that yardstick where it exceeds 1e-4.
    tol_col = max(TOL, 3.0 * float(g[f"yard_col_{variant}"]))
    print(f"    colour tolerance {tol_col:.2e} (the reference's own fp32-vs-float64 distance: {float(g[f'yard_col_{variant}']):.2e})")
    assert e_sig < TOL and e_d < TOL and e_col < tol_col and e_rgb < tol_col
    assert n_bad <= 4
    np.testing.assert_allclose(wts.cpu().sum(-1)[ok].numpy(), g[f"weights_sum_{variant}"][ok.numpy()], rtol=1e-4, atol=3e-5)
    if precision == "f16x3":
        if variant == "a":
            assert fb == 0, f"the f16x3 kernels fell back on realistic magnitudes ({fb} launches): the headline mode is not what such a checkpoint runs in"
        else:
            assert fb >= 1, "activations beyond the fp16 range did not raise the range flag inside a render"
    else:
        assert fb == 0
