"""Row f1 (training path): the HIP GEMM building block against torch, and the gradients of the radiance field and the
compositor against torch autograd on the CPU oracle (the form in which the reference itself differentiates them,
diner.py:217-290).  Tolerance: 1e-4 max-norm relative per gradient tensor, as for the forward path."""
import numpy as np
import os

import pytest
import torch

from tests.helpers import load, oracle_setup, max_norm_rel
from oracle import diner_oracle as O

pytestmark = pytest.mark.gpu


def T(a):
    return torch.as_tensor(np.asarray(a)).float()
TOL_GRAD = 1e-4


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    from diner_amd import ops as _ops
    return _ops


def test_gemm_against_torch(ops):
    from diner_amd import train
    g = torch.Generator().manual_seed(1)
    # (300,512,55) / (129,65,17) / (1000,4,512) / (4,512,2500): fp32-MFMA kernels and ragged edges; the others are layer-sized and
    # take the split-bf16 ("bf16x6") kernel, including ragged M / N / K tiles and operand values spanning 1e-8 .. 1e4
    for (M, N, K) in ((300, 512, 55), (1000, 4, 512), (4, 512, 2500), (129, 65, 17), (512, 512, 4096), (1000, 200, 70),
                      (20480, 512, 512)):
        A = torch.randn(M, K, generator=g); B = torch.randn(K, N, generator=g)
        if (M, N, K) == (1000, 200, 70):      # wide dynamic range: a loss gradient next to activations (bf16 keeps fp32's exponent)
            A = A * torch.logspace(-8, 4, M).unsqueeze(1)
            A = A / A.abs().max()
        exact_ref = None
        bias = torch.randn(N, generator=g); mask = torch.randn(M, N, generator=g)
        C0 = torch.randn(M, N, generator=g)
        ref = A.double() @ B.double()
        Ac, Bc = A.cuda(), B.cuda()
        At, Bt = A.t().contiguous().cuda(), B.t().contiguous().cuda()
        scale = ref.abs().max().item()

        def run(a, b, lda, ldb, flags, **kw):
            C = C0.clone().cuda()
            train.gemm(a, b, C, M, N, K, lda, ldb, N, flags, **kw)
            return C.cpu().double()
        assert (run(Ac, Bc, K, N, 0) - ref).abs().max() / scale < 1e-5
        assert (run(At, Bc, M, N, train.TA) - ref).abs().max() / scale < 1e-5
        assert (run(Ac, Bt, K, K, train.TB) - ref).abs().max() / scale < 1e-5
        assert (run(At, Bt, M, K, train.TA | train.TB) - ref).abs().max() / scale < 1e-5
        assert (run(Ac, Bc, K, N, train.ACCUM, bias=bias.cuda()) - (ref + bias.double() + C0.double())).abs().max() / scale < 1e-5
        want = (A.clamp(min=0).double() @ B.clamp(min=0).double()) * (mask > 0).double()
        assert (run(Ac, Bc, K, N, train.RELU_A | train.RELU_B, mask=mask.cuda()) - want).abs().max() / want.abs().max() < 1e-5
        C = torch.zeros(M, N).cuda()
        train.gemm(Ac, Bc, C, M, N, K, K, N, N, train.ATOMIC, k_split=7)
        assert (C.cpu().double() - ref).abs().max() / scale < 1e-5
        # the exact-fp32 flag selects the fp32 MFMA for every shape; both arithmetic paths are fp32-class
        e_x = ((run(Ac, Bc, K, N, train.EXACT) - ref).abs().max() / scale).item()
        e_b = ((run(Ac, Bc, K, N, 0) - ref).abs().max() / scale).item()
        print(f"gemm {M}x{N}x{K}: max-norm error vs float64: exact-fp32 MFMA {e_x:.1e}, default path {e_b:.1e}")
        assert e_x < 1e-5 and e_b < 1e-5
    with pytest.raises(RuntimeError):
        train.gemm(Ac, Bc, C, M, N, K, K, N, N, 0, k_split=2)             # split-K without the atomic flag


def _oracle_grads(scene, w, xyz, dirs, G, relu_masks=None):
    scene.latent.requires_grad_(True)
    scene.latent.grad = None
    names = []
    for k, v in vars(w).items():
        for i, t in enumerate(v if isinstance(v, (list, tuple)) else [v]):
            if torch.is_tensor(t) and t.is_floating_point():
                t.requires_grad_(True)
                t.grad = None
                names.append((k, i if isinstance(v, (list, tuple)) else None, t))
    out = O.pixelnerf_forward(scene, w, xyz, dirs, relu_masks)
    (out * G).sum().backward()
    return out.detach(), scene.latent.grad, {(k, i): t.grad for k, i, t in names}


@pytest.mark.parametrize("P,fused", [(200, False), (200, True), (5120, False), (5120, True), (20480, True)],
                         ids=["200-layerwise", "200-fused_forward", "5120-layerwise", "5120-fused_forward", "20480-fused_forward"])
def test_field_forward_and_backward_against_oracle_autograd(ops, P, fused, monkeypatch):
    """P = 20480 (round 5): 512 rays x 40 samples = 81920 rows per view layer -- whole rounds of 128-row tiles + a
    ragged rest in the f16x3 backward, the 256 x 256 weight-gradient tiles over 64 row chunks and the fused forward at a size where every
    CU owns several tiles, against the ORACLE's autograd (until now these launch shapes were tied to it only through HIP-vs-HIP sums).
    P = 200: 800 / 200 rows per layer (the general kernel serves the post-mean layers).  P = 5120: the reference training batch's row
    counts (128 rays x 40 samples: 20480 rows per view layer, 5120 behind the view mean) -- the launch plans of k_run512 that the timing
    runs use: one round of 64-row tiles + shared 32-row tiles, the weight-gradient product in the same launch, the deferred summing pass."""
    from diner_amd import train
    from tests.tests_train_util import module_param_list
    # fused_forward (round 5, DINER_TRAIN_FUSED_FWD=1): the forward on the storing variants of the inference kernels (k_train_fwd_pre /
    # k_train_fwd_post); the backward reads what they saved -- the same bars hold, including the one conditioned on the saved relu decisions
    monkeypatch.setenv("DINER_TRAIN_FUSED_FWD", "1" if fused else "0")
    g = load("g6_pixelnerf.npz")
    sc, scene, w, msd, rays = oracle_setup(int(g["W"]), int(g["H"]), int(g["seed"]))
    reps = (P + g["pts"].shape[0] - 1) // g["pts"].shape[0]
    jit = 1.0 + 1e-3 * torch.arange(reps).repeat_interleave(g["pts"].shape[0])[:P, None]      # the repeats are not exact copies
    xyz, dirs = T(g["pts"]).repeat(reps, 1)[:P] * jit, T(g["dirs"]).repeat(reps, 1)[:P]
    G = torch.randn(P, 4, generator=torch.Generator().manual_seed(3))
    out_o, dlat_o, gr_o = _oracle_grads(scene, w, xyz, dirs, G)
    hs = ops.HipScene(sc["latent"].detach().cuda(), sc["depths"].cuda(), sc["depths_std"].cuda(), sc["normals"].cuda(),
                      sc["src_extrinsics"], sc["src_intrinsics"][:, [0, 1], [0, 1]], sc["src_intrinsics"][:, :2, -1],
                      sc["image_shape"], sc["feature_padding"])
    latent = sc["latent"].detach().cuda().requires_grad_(True)
    params, names = module_param_list(msd)
    out = train.field_train(hs, xyz.cuda(), dirs.cuda(), latent, params)
    e_fwd = max_norm_rel(out.detach().cpu(), out_o)
    (out * G.cuda()).sum().backward(retain_graph=True)
    e_lat = max_norm_rel(latent.grad.cpu(), dlat_o)
    worst = ("", 0.0)
    for p, (k, i) in zip(params, names):
        e = max_norm_rel(p.grad.cpu(), gr_o[(k, i)])
        if e > worst[1]:
            worst = (f"{k}[{i}]", e)
    print(f"training path: forward {e_fwd:.2e}, d latent {e_lat:.2e}, worst parameter gradient {worst[0]} {worst[1]:.2e}")
    if P <= 200:
        assert e_fwd < 2e-5 and e_lat < TOL_GRAD and worst[1] < TOL_GRAD
    else:
        # 20480 rows x 512 units x 13 layers hold a few hundred pre-activations within rounding of zero: two fp32 evaluations (and an fp32
        # and an fp64 one: the oracle in float64 is 1.4e-2 from the oracle in float32 in this norm) put some of them on different sides of
        # the relu, and one flipped unit of one row moves a whole row of a weight gradient by one of the ~20480 random-sign terms its
        # entries are sums of -- ~1e-2 of the largest entry.  The flips are few: in the Frobenius norm the gradients agree.
        rms = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
        r_lat = rms(latent.grad.cpu(), dlat_o)
        r_par = max(rms(p.grad.cpu(), gr_o[(k, i)]) for p, (k, i) in zip(params, names))
        print(f"   Frobenius-relative: d latent {r_lat:.2e}, worst parameter gradient {r_par:.2e}")
        assert e_fwd < 2e-5 and e_lat < 3e-2 and worst[1] < 3e-2 and r_lat < 2e-3 and r_par < 2e-3
        # The sharp statement (round 4): the reference's backward CONDITIONED on the relu decisions of the HIP forward (signs of the
        # pre-activations it saved, diner_field_train_ws_layout) -- a flipped relu is no longer a difference, so a wrong row range of a
        # launch plan, a dropped tile or a mis-scaled operand cannot hide behind the flips: max-norm 1e-4 on every gradient tensor.
        from tests.tests_train_util import saved_relu_masks
        masks = saved_relu_masks(out, P)
        out_c, dlat_c, gr_c = _oracle_grads(scene, w, xyz, dirs, G, masks)
        e_fwd_c = max_norm_rel(out.detach().cpu(), out_c)
        e_lat_c = max_norm_rel(latent.grad.cpu(), dlat_c)
        worst_c = max(((f"{k}[{i}]", max_norm_rel(p.grad.cpu(), gr_c[(k, i)])) for p, (k, i) in zip(params, names)), key=lambda t: t[1])
        print(f"   conditioned on the HIP forward's relu decisions: forward {e_fwd_c:.2e}, d latent {e_lat_c:.2e}, worst parameter gradient "
              f"{worst_c[0]} {worst_c[1]:.2e}")
        assert e_fwd_c < 2e-5 and e_lat_c < TOL_GRAD and worst_c[1] < TOL_GRAD
    assert (latent.grad != 0).any() and all((p.grad != 0).any() for p in params)
    # a second backward through the retained graph (the gradient buffers of the first one, allocated during the forward, are the
    # parameters' .grad by now and must not be written again): everything doubles
    first = [p.grad.clone() for p in params] + [latent.grad.clone()]
    (out * G.cuda()).sum().backward()
    for a, b in zip(first, [p.grad for p in params] + [latent.grad]):
        assert max_norm_rel(b.cpu(), 2 * a.cpu()) < 1e-5


@pytest.mark.parametrize("scale", [1e-6, 1e-9])
def test_small_batch_backward_with_small_upstream_gradients(ops, scale):
    """with P < 256 the post-mean layers (M = P rows) leave the 512 x 512 kernels, and until round 5 the general-GEMM
    fall-back wrote no maximum of its dx, so blocks 2..0 (M = P nv >= 256, f16x3) staged their dy operand UNSCALED -- harmless for the O(1)
    upstream gradients of the test above, but loss gradients of 1e-6 .. 1e-9 lost most of their bits in two fp16 planes.  The backward is
    linear in the upstream gradient: the gradients for G x scale must be scale x the gradients for G (pinned against the oracle's autograd
    by test_field_forward_and_backward_against_oracle_autograd[200]) to fp32 round-off."""
    from diner_amd import train
    from tests.tests_train_util import module_param_list
    P = 200
    g = load("g6_pixelnerf.npz")
    sc, scene, w, msd, rays = oracle_setup(int(g["W"]), int(g["H"]), int(g["seed"]))
    xyz, dirs = T(g["pts"])[:P], T(g["dirs"])[:P]
    G = torch.randn(P, 4, generator=torch.Generator().manual_seed(3))
    hs = ops.HipScene(sc["latent"].detach().cuda(), sc["depths"].cuda(), sc["depths_std"].cuda(), sc["normals"].cuda(),
                      sc["src_extrinsics"], sc["src_intrinsics"][:, [0, 1], [0, 1]], sc["src_intrinsics"][:, :2, -1],
                      sc["image_shape"], sc["feature_padding"])
    grads = {}
    for s in (1.0, scale):
        latent = sc["latent"].detach().cuda().requires_grad_(True)
        params, names = module_param_list(msd)
        out = train.field_train(hs, xyz.cuda(), dirs.cuda(), latent, params)
        (out * (G * s).cuda()).sum().backward()
        grads[s] = [p.grad.cpu() for p in params] + [latent.grad.cpu()]
    out_o, dlat_o, gr_o = _oracle_grads(scene, w, xyz, dirs, G * scale)
    worst = max(max_norm_rel(b / scale, a) for a, b in zip(grads[1.0], grads[scale]))
    worst_o = max([max_norm_rel(b, gr_o[(k, i)]) for b, (k, i) in zip(grads[scale], names)] + [max_norm_rel(grads[scale][-1], dlat_o)])
    print(f"upstream gradients x {scale:g}: HIP gradients against {scale:g} x the O(1) run {worst:.2e}, against the oracle's autograd {worst_o:.2e}")
    assert worst < 2e-5 and worst_o < TOL_GRAD


def test_composite_backward_against_oracle_autograd(ops):
    from diner_amd import train
    g = torch.Generator().manual_seed(11)
    NR, K = 70, 40
    field = torch.rand(NR, K, 4, generator=g)
    field[..., 3] = torch.relu(torch.randn(NR, K, generator=g)) * 30
    rays = torch.zeros(NR, 8); rays[:, 6] = 0.5; rays[:, 7] = 1.5
    z = (0.5 + torch.rand(NR, K, generator=g)).sort(-1).values.clamp(max=1.49)
    z[3, -1] = 1.6                                                     # a sample beyond `far`: negative delta (:301)
    Grgb, Gd = torch.randn(NR, 3, generator=g), torch.randn(NR, generator=g)
    for white in (False, True):
        fo = field.clone().requires_grad_(True)
        _, rgb_o, d_o = O.composite_from_field(fo, rays, z, white)
        ((rgb_o * Grgb).sum() + (d_o * Gd).sum()).backward()
        fh = field.clone().cuda().requires_grad_(True)
        rgb, dep = train.composite_train(fh, z.cuda(), rays.cuda(), white)
        assert max_norm_rel(rgb.detach().cpu(), rgb_o.detach()) < 1e-5
        ((rgb * Grgb.cuda()).sum() + (dep * Gd.cuda()).sum()).backward()
        e = max_norm_rel(fh.grad.cpu(), fo.grad)
        print(f"compositor adjoint (white={white}): {e:.2e}")
        assert e < TOL_GRAD


def test_module_training_step_against_oracle_autograd(ops):
    """The drop-in modules in grad mode, as DINER.calc_losses drives them (diner.py:259-266: renderer.forward on a ray
    batch, loss on fine.rgb): same rgb as the inference path, and the gradients of a rgb loss with respect to every MLP
    parameter and to encoder.latent match torch autograd through the CPU oracle on the same sample positions."""
    from tests.test_boundary_gpu import setup_model
    from diner_amd import noise, train
    sc, nerf, R, rays = setup_model(32, 32, 4)
    nerf.train()
    NR, K, G, n_cand = 96, 40, 15, 1000
    sel = torch.linspace(0, rays.shape[0] - 1, NR).long()
    r = rays[sel].cuda()[None]
    gen = torch.Generator().manual_seed(21)
    inj = (torch.rand(1, NR, n_cand, generator=gen).cuda(), torch.randn(1, NR, G, generator=gen).cuda(),
           torch.rand(1, NR, K, generator=gen).cuda())
    ren = R(n_samples=K, n_depth_candidates=n_cand, n_gaussian=G, white_bkgd=True)
    nerf.encoder.latent = nerf.encoder.latent.detach().requires_grad_(True)
    with noise.inject(*inj):
        with torch.no_grad():
            z = ren.fill_up_uniform_samples(ren.sample_depthguided(r, nerf, K, n_cand, n_gaussian=G), r)
            ref_out = ren.forward(nerf, r).fine.rgb                    # inference kernels, same noise -> same z
        assert nerf.needs_grad()
        out = ren.forward(nerf, r)
    assert out.fine.rgb.requires_grad
    assert max_norm_rel(out.fine.rgb.detach().cpu(), ref_out.cpu()) < 2e-5
    Gm = torch.randn(1, NR, 3, generator=gen)
    (out.fine.rgb * Gm.cuda()).sum().backward()
    # oracle: same z, torch autograd on the CPU
    _, scene, w, msd, _ = oracle_setup(32, 32, 4)
    scene.latent.requires_grad_(True)
    leaves = {}
    for k, v in vars(w).items():
        for i, t in enumerate(v if isinstance(v, (list, tuple)) else [v]):
            if torch.is_tensor(t) and t.is_floating_point():
                leaves[(k, i if isinstance(v, (list, tuple)) else None)] = t.requires_grad_(True)
    rc, zc = r[0].cpu(), z[0].cpu()
    xyz = (rc[:, None, :3] + zc[..., None] * rc[:, None, 3:6]).reshape(-1, 3)
    dirs = rc[:, None, 3:6].expand(-1, K, -1).reshape(-1, 3)
    field = O.pixelnerf_forward(scene, w, xyz, dirs).view(NR, K, 4)
    _, rgb_o, _ = O.composite_from_field(field, rc, zc, True)
    (rgb_o * Gm[0]).sum().backward()
    assert max_norm_rel(out.fine.rgb[0].detach().cpu(), rgb_o.detach()) < 2e-5
    from tests.tests_train_util import oracle_key
    # float64 run of the same oracle: tells how far float32 autograd itself is from the exact gradient
    import copy
    scene64 = copy.copy(scene)
    for k, v in vars(scene).items():
        if torch.is_tensor(v) and v.is_floating_point():
            setattr(scene64, k, v.detach().double())
    scene64.latent.requires_grad_(True)
    w64 = copy.copy(w)
    leaves64 = {}
    for k, v in vars(w).items():
        if isinstance(v, (list, tuple)):
            new = [t.detach().double().requires_grad_(True) for t in v]
            setattr(w64, k, new)
            for i, t in enumerate(new):
                leaves64[(k, i)] = t
        elif torch.is_tensor(v) and v.is_floating_point():
            t = v.detach().double().requires_grad_(True)
            setattr(w64, k, t)
            leaves64[(k, None)] = t
    f64 = O.pixelnerf_forward(scene64, w64, xyz.double(), dirs.double()).view(NR, K, 4)
    _, rgb64, _ = O.composite_from_field(f64, rc.double(), zc.double(), True)
    (rgb64 * Gm[0].double()).sum().backward()
    worst, worst_o = ("", 0.0), ("", 0.0)
    for name, p in nerf.mlp_fine.named_parameters():
        exact = leaves64[oracle_key(name)].grad
        worst = max(worst, (name, max_norm_rel(p.grad.cpu().double(), exact)), key=lambda t: t[1])
        worst_o = max(worst_o, (name, max_norm_rel(leaves[oracle_key(name)].grad.double(), exact)), key=lambda t: t[1])
    e_lat = max_norm_rel(nerf.encoder.latent.grad[0].cpu().double(), scene64.latent.grad)
    e_lat_o = max_norm_rel(scene.latent.grad.double(), scene64.latent.grad)
    print(f"module training step vs float64 autograd: HIP worst parameter gradient {worst[0]} {worst[1]:.2e}, d latent "
          f"{e_lat:.2e}; float32 torch autograd itself: {worst_o[0]} {worst_o[1]:.2e}, d latent {e_lat_o:.2e}")
    # a rgb loss through 40 alpha-composited samples is ill-conditioned in float32: float32 torch autograd itself is
    # ~2e-3 from the float64 gradient on the worst tensor (pre-activations within rounding of zero land on either side of the relu, and
    # one flipped unit of one row is one of the few thousand random-sign terms an entry of a weight gradient sums); the HIP path -- another
    # fp32-class evaluation with its own roundings, hence its own flips -- must be in that class (twice fp32 autograd's own distance) and
    # at 1e-4 when it is easy
    assert worst[1] < max(TOL_GRAD, 2.0 * worst_o[1]) and e_lat < max(TOL_GRAD, 2.0 * e_lat_o)


def test_three_adam_steps_track_torch_autograd(ops):
    """A miniature of Trainer.fit on the MLP (diner.py:292-299 optimises with Adam): three optimisation steps of a rgb
    loss through renderer.composite on fixed sample positions, once with the drop-in modules on the HIP training path
    and once with torch autograd through the CPU oracle; the loss trajectories must agree."""
    from tests.test_boundary_gpu import setup_model
    from tests.tests_train_util import oracle_key
    sc, nerf, R, rays = setup_model(32, 32, 6)
    nerf.train()
    NR, K = 64, 40
    r = rays[torch.linspace(0, rays.shape[0] - 1, NR).long()]
    gen = torch.Generator().manual_seed(33)
    z = (r[:, 6:7] + (r[:, 7:8] - r[:, 6:7]) * torch.rand(NR, K, generator=gen).sort(-1).values)
    target = torch.rand(NR, 3, generator=gen)
    ren = R(n_samples=K, n_depth_candidates=1000, n_gaussian=15, white_bkgd=True)
    opt = torch.optim.Adam(nerf.mlp_fine.parameters(), lr=1e-3)
    losses = []
    for _ in range(3):
        opt.zero_grad()
        _, rgb, _ = ren.composite(nerf, r.cuda()[None], z.cuda()[None])
        loss = (rgb[0] - target.cuda()).square().mean()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    # the same three steps with torch autograd on the CPU oracle
    _, scene, w, msd, _ = oracle_setup(32, 32, 6)
    leaves = {}
    for k, v in vars(w).items():
        for i, t in enumerate(v if isinstance(v, (list, tuple)) else [v]):
            if torch.is_tensor(t) and t.is_floating_point():
                leaves[(k, i if isinstance(v, (list, tuple)) else None)] = t.requires_grad_(True)
    names = [n for n, _ in nerf.mlp_fine.named_parameters()]
    opt_o = torch.optim.Adam([leaves[oracle_key(n)] for n in names], lr=1e-3)
    xyz = (r[:, None, :3] + z[..., None] * r[:, None, 3:6]).reshape(-1, 3)
    dirs = r[:, None, 3:6].expand(-1, K, -1).reshape(-1, 3)
    losses_o = []
    for _ in range(3):
        opt_o.zero_grad()
        field = O.pixelnerf_forward(scene, w, xyz, dirs).view(NR, K, 4)
        _, rgb_o, _ = O.composite_from_field(field, r, z, True)
        loss = (rgb_o - target).square().mean()
        loss.backward()
        opt_o.step()
        losses_o.append(loss.item())
    print("losses HIP   ", " ".join(f"{l:.6f}" for l in losses))
    print("losses oracle", " ".join(f"{l:.6f}" for l in losses_o))
    assert max(abs(a - b) / b for a, b in zip(losses, losses_o)) < 1e-4
    worst = max(max_norm_rel(p.detach().cpu(), leaves[oracle_key(n)].detach()) for n, p in nerf.mlp_fine.named_parameters())
    print(f"weights after 3 Adam steps: worst max-norm relative difference {worst:.2e}")
    # (informational: Adam's first steps move every weight by +-lr whatever the gradient's size, so weights whose
    #  gradient is ~0 can step in opposite directions in the two runs; the loss trajectory is the meaningful check)


def test_gradient_reaches_the_image_encoder(ops):
    """encode() in grad mode (ResNet trunk in torch) -> renderer.forward on the HIP training path -> loss.backward():
    the latent gradient returned by the HIP backward continues through torch autograd into the encoder's convolutions, as
    in the reference's training (train.py -> DINER.training_step)."""
    from tests.test_boundary_cpu import build_nerf
    from diner_amd.synthetic import make_scene, make_mlp_state_dict
    from src.util.import_helper import import_obj
    from src.util.cam_geometry import gen_rays
    W = H = 32
    sc = make_scene(W, H, seed=4, latent=False)
    nerf = build_nerf().cuda().train()
    nerf.mlp_fine.load_state_dict(make_mlp_state_dict())
    imgs = torch.rand(1, 4, 3, H, W, generator=torch.Generator().manual_seed(0)).cuda()
    nerf.encode(imgs, sc["depths"][None].cuda(), sc["depths_std"][None].cuda(), sc["src_extrinsics"][None].cuda(),
                sc["src_intrinsics"][None].cuda())
    assert nerf.encoder.latent.requires_grad
    from diner_amd import train as _tr
    assert _tr._channels_last(nerf.encoder.latent)      # round 6: the pyramid is concatenated channels-last on a HIP device; its gradient returns as a view in that format
    rays = gen_rays(sc["target_extrinsics"].view(1, 4, 4).cuda(), sc["target_intrinsics"].view(1, 3, 3).cuda(), W, H,
                    torch.tensor([sc["znear"]]).cuda(), torch.tensor([sc["zfar"]]).cuda()).view(1, -1, 8)
    ren = import_obj("src.models.nerf_renderer.NeRFRendererDGS")(n_samples=40, n_gaussian=15, white_bkgd=True)
    out = ren.forward(nerf, rays[:, ::8].contiguous())
    (out.fine.rgb - 0.5).square().mean().backward()
    conv = [p for n, p in nerf.encoder.named_parameters() if n.endswith("conv1.weight")][0]
    assert conv.grad is not None and torch.isfinite(conv.grad).all() and float(conv.grad.abs().max()) > 0
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in nerf.mlp_fine.parameters())


def test_linear512_against_float64():
    """The feature-sliced 512 x 512 training product (csrc/train_lin512.hip) against float64: forward y = relu(x) W^T + b + resid and the
    data gradient dx = (dy W) * (mask > 0) accumulated onto dx, ragged row counts (1, 63, 65, 1000: partial tiles, fewer tiles than CUs;
    12289: 32-row tiles shared by two workgroups; 20480: one round of 64-row tiles + shared 32-row tiles; 70000: 64-row rounds + 32-row tiles), operands spanning 1e-8 .. 1e4 in magnitude (bf16 planes keep fp32's exponent range)."""
    from diner_amd import train
    g = torch.Generator().manual_seed(3)
    W = (torch.randn(512, 512, generator=g) * 0.05).cuda()
    b = torch.randn(512, generator=g).cuda()
    for M in (1, 63, 65, 1000, 12289, 20480, 70000):
        x = torch.randn(M, 512, generator=g).cuda() * torch.logspace(-8, 4, 512).cuda()[torch.randperm(512, generator=g).cuda()]
        r = torch.randn(M, 512, generator=g).cuda()
        y = torch.empty(M, 512, device="cuda")
        train.linear512(x, W, y, relu_in=True, bias=b, resid=r)
        want = (torch.relu(x.double()) @ W.double().T + b.double() + r.double())
        scale = (torch.relu(x.double()).abs() @ W.double().abs().T).max()
        err = float((y.double() - want).abs().max() / scale)
        # data gradient with mask and accumulation
        dy = torch.randn(M, 512, generator=g).cuda() * 1e-6
        m = torch.randn(M, 512, generator=g).cuda()
        dx = torch.randn(M, 512, generator=g).cuda() * 1e-6
        want_dx = dx.double() + (dy.double() @ W.double()) * (m.double() > 0)
        train.linear512(dy, W, dx, transpose=True, accumulate=True, mask=m)
        scale2 = (dy.double().abs() @ W.double().abs()).max()
        err2 = float((dx.double() - want_dx).abs().max() / scale2)
        print(f"linear512 M={M}: forward {err:.2e}, dgrad {err2:.2e} (max error / max sum of |products|)")
        assert err < 1e-6 and err2 < 1e-6
        # the forward product in the f16x3 arithmetic (what the training forward runs first): operands inside the fp16 range
        xs = torch.randn(M, 512, generator=g).cuda() * torch.logspace(-6, 3, 512).cuda()[torch.randperm(512, generator=g).cuda()]
        train.linear512(xs, W, y, relu_in=True, bias=b, resid=r, f16x3=True)
        want3 = torch.relu(xs.double()) @ W.double().T + b.double() + r.double()
        err3 = float((y.double() - want3).abs().max() / (torch.relu(xs.double()).abs() @ W.double().abs().T).max())
        print(f"   f16x3 forward {err3:.2e}")
        assert err3 < 2e-6


def test_wgrad512_against_float64():
    """The persistent weight-gradient kernel (csrc/train_wgrad512.hip) against float64: dW = dy^T relu(x), db = column sums of dy, ragged
    row counts (fewer rows than a slab, partial slabs, 1 .. 32 row chunks), operands spanning 1e-8 .. 1e4."""
    from diner_amd import train
    g = torch.Generator().manual_seed(4)
    for M in (1, 31, 257, 4097, 20480, 70001):
        x = torch.randn(M, 512, generator=g).cuda() * torch.logspace(-6, 3, 512).cuda()[torch.randperm(512, generator=g).cuda()]
        dy = torch.randn(M, 512, generator=g).cuda() * torch.logspace(-8, 0, 512).cuda()[torch.randperm(512, generator=g).cuda()]
        want = dy.double().T @ torch.relu(x.double())
        scale = (dy.double().abs().T @ torch.relu(x.double()).abs()).max()
        want_b = dy.double().sum(0)
        scratch = torch.full((train.lib.diner_wgrad512_scratch_bytes(),), 0xFF, dtype=torch.uint8, device="cuda")   # NaN patterns: stale reads show
        for how, ws in (("atomics", None), ("partials", scratch)):
            dW = torch.zeros(512, 512, device="cuda")
            db = torch.zeros(512, device="cuda")
            train.wgrad512(dy, x, dW, db, relu_in=True, scratch=ws)
            e_w = float((dW.double() - want).abs().max() / scale)
            e_b = float((db.double() - want_b).abs().max() / dy.double().abs().sum(0).max())
            print(f"wgrad512 M={M} ({how}): dW {e_w:.2e}, db {e_b:.2e} (max error / max sum of |products|)")
            assert e_w < 2e-6 and e_b < 2e-6


def test_scatter_latent_merged_and_layout_pass_against_torch():
    """The two passes the latent gradient leaves through (csrc/train.hip), through their stage entries, against torch index_add / permute:
    k_scatter_latent_merged on taps that merge (consecutive columns on neighbouring texels), on taps that do not (every tap its own texel:
    the 56-slot table overflows and the rest goes out as direct atomics), with zero weights and a column count that is no multiple of 64;
    k_cl_to_nchw on map sizes and channel counts that are no multiples of its 64 x 64 tile."""
    from diner_amd import train, _lib
    from diner_amd.ops import _ptr, _stream
    lib = train.lib
    g = torch.Generator().manual_seed(11)
    n_tex = 4 * 24 * 24
    for name, cols, rows in (("merging", 1000, None), ("distinct", 333, "distinct"), ("one texel", 130, "one")):
        if rows is None:          # a walk over the texels: consecutive columns share most of their taps
            base = (torch.arange(cols) // 3) % (n_tex - 30)
            tap_row = torch.stack([base, base + 1, base + 24, base + 25], 1).int()
        elif rows == "distinct":
            tap_row = torch.randperm(n_tex, generator=g)[:cols * 4].view(cols, 4).int()
        else:
            tap_row = torch.full((cols, 4), 77, dtype=torch.int32)
        tap_w = torch.rand(cols, 4, generator=g)
        tap_w[torch.rand(cols, 4, generator=g) < 0.2] = 0.0
        d_lat = torch.randn(cols, 512, generator=g)
        want = torch.zeros(n_tex, 512, dtype=torch.float64)
        for k in range(4):
            want.index_add_(0, tap_row[:, k].long(), d_lat.double() * tap_w[:, k:k + 1].double())
        out = torch.zeros(n_tex, 512, device="cuda")
        dl, tr, tw = d_lat.cuda(), tap_row.cuda(), tap_w.cuda()
        _lib.check(lib.diner_scatter_latent_grad_f32(_ptr(dl), _ptr(tr), _ptr(tw), cols, _ptr(out), _stream()))
        err = float((out.cpu().double() - want).abs().max() / want.abs().max())
        print(f"scatter ({name}, {cols} columns): {err:.2e}")
        assert err < 1e-5
    for n, H, W, C in ((1, 5, 7, 3), (2, 24, 24, 512), (3, 9, 33, 70)):
        src = torch.randn(n, H, W, C, generator=g).cuda()
        dst = torch.empty(n, C, H, W, device="cuda")
        _lib.check(lib.diner_channels_last_to_nchw_f32(_ptr(src), n, H * W, C, _ptr(dst), _stream()))
        assert torch.equal(dst, src.permute(0, 3, 1, 2).contiguous()), (n, H, W, C)


@pytest.mark.parametrize("fused", [False, True, "batched"], ids=["layerwise", "fused_forward", "fused_forward_batched_node"])
def test_training_forward_beyond_the_fp16_range(ops, fused, monkeypatch):
    """fused_forward (the default since round 5): the storing inference kernels raise their flag and the host repeats the object on the
    layer-wise forward.  The layer-wise training forward runs its 512 x 512 products in the f16x3 arithmetic first; an operand beyond the fp16 range raises the product's
    flag and the bf16x6 launch behind it recomputes it (no host synchronisation).  One feature of the residual stream sits at 1e5 (the
    f16x3 products that stage it would turn it into inf and their outputs into NaN): outputs and gradients follow the oracle's autograd."""
    from diner_amd import train
    from tests.tests_train_util import module_param_list
    monkeypatch.setenv("DINER_TRAIN_FUSED_FWD", "1" if fused else "0")
    g = load("g6_pixelnerf.npz")
    sc, scene, w, msd, rays = oracle_setup(int(g["W"]), int(g["H"]), int(g["seed"]))
    big = {k: v.clone() for k, v in msd.items()}
    big["lin_in.bias"][5] += 1.0e5                      # one feature of the residual stream beyond 65504 in every block ...
    for k in big:                                       # ... that nothing reads with a non-zero weight: the outputs stay ordinary
        if k.endswith("fc_0.weight") or k == "lin_out.weight":
            big[k][:, 5] = 0.0
    wb = O.MLPWeights.from_state_dict(big)
    P = 300
    xyz, dirs = T(g["pts"])[:P], T(g["dirs"])[:P]
    G = torch.randn(P, 4, generator=torch.Generator().manual_seed(5))
    out_o, dlat_o, gr_o = _oracle_grads(scene, wb, xyz, dirs, G)
    hs = ops.HipScene(sc["latent"].detach().cuda(), sc["depths"].cuda(), sc["depths_std"].cuda(), sc["normals"].cuda(),
                      sc["src_extrinsics"], sc["src_intrinsics"][:, [0, 1], [0, 1]], sc["src_intrinsics"][:, :2, -1],
                      sc["image_shape"], sc["feature_padding"])
    latent = sc["latent"].detach().cuda().requires_grad_(True)
    params, names = module_param_list(big)
    if fused == "batched":      # round 6: the one-node path skips the gather of the interpolated latent rows -- the gated repeat gathers them behind
        out = train.field_train_batch([hs], xyz.cuda()[None], dirs.cuda()[None], latent[None], params)[0]      # its gate, the backward's sample-space
        if os.environ.get("DINER_TRAIN_BATCH", "") != "0":                                                              # lin_z adjoint (300 points) regathers
            assert "Batch" in type(out.grad_fn).__name__ or "Batch" in type(out.grad_fn.next_functions[0][0]).__name__
    else:
        out = train.field_train(hs, xyz.cuda(), dirs.cuda(), latent, params)
    assert torch.isfinite(out).all()
    if fused is True:       # the fused kernels did raise their flag: what is compared below is the gated layer-wise repeat
        import ctypes as C
        from diner_amd import _lib
        ovf = C.c_int(0)
        _lib.check(_lib.load().diner_field_train_fused_overflowed(out.grad_fn.saved_tensors[0].data_ptr(), P, 4, C.byref(ovf), None))
        assert ovf.value == 1
    e_fwd = float((out.detach().cpu() - out_o).abs().max())          # sigmoid / relu outputs of O(1) .. O(1e5): absolute on rgb, relative on sigma
    (out * G.cuda()).sum().backward()
    worst = 0.0
    for p, (k, i) in zip(params, names):
        want = gr_o[(k, i)]
        assert torch.isfinite(p.grad).all()
        if float(want.abs().max()) == 0.0:
            assert float(p.grad.abs().max()) == 0.0, (k, i)
        else:
            worst = max(worst, max_norm_rel(p.grad.cpu(), want))
    print(f"beyond the fp16 range: forward max abs difference {e_fwd:.2e} (sigma up to {float(out_o[:, 3].max()):.3g}), worst parameter gradient {worst:.2e}")
    assert max_norm_rel(out.detach().cpu(), out_o) < 2e-5 and worst < TOL_GRAD


def _field_nodes(t):
    """autograd nodes of diner_amd.train.FieldFunction under tensor t, in the order of the objects of the batch (torch.stack keeps it)."""
    seen, out, stack = set(), [], [t.grad_fn]
    while stack:
        n = stack.pop(0)
        if n is None or n in seen:
            continue
        seen.add(n)
        if "FieldFunction" in type(n).__name__ or "FieldBatchFunction" in type(n).__name__:
            out.append(n)
            continue
        stack.extend(f for f, _ in n.next_functions)
    return out


@pytest.mark.parametrize("batched", [True, False], ids=["one_call_pair_for_the_objects", "one_call_pair_per_object"])
def test_training_step_two_objects_patch_of_rays(ops, monkeypatch, batched):
    """(round 6) batched: the SB objects as ONE field node (ABI v6, diner_field_train_forward_batch_f32 / _backward_batch_f32: the layer
    products of the backward run once over all objects' rows, weight gradients summed in-kernel); else DINER_TRAIN_BATCH=0, round 5's one
    call pair per object with autograd summing the gradient sets.
    The step the shipped configs run, in small (DINER.calc_losses, diner.py:217-290 with configs/train_dtu.yaml:16,63: SB objects, a square
    patch of rays per object, ONE renderer.forward on (SB, B, 8) rays in grad mode): SB = 2 objects with their own source views and
    feature maps, a 32 x 32 patch each (the shipped patch is 64 x 64 = 4096 rays; the CPU oracle's saved activations are 10 GB per object at
    32 x 32 already), loss on fine.rgb.  Statements: per-object rgb as the inference path; gradients of every MLP parameter -- SUMMED over the
    objects by autograd across the per-object gradient sets -- and of each object's slab of encoder.latent against torch autograd through
    the CPU oracle: in the Frobenius norm unconditioned, at 1e-4 max-norm conditioned on the HIP forward's relu decisions."""
    from diner_amd import noise
    from diner_amd.synthetic import make_scene, make_mlp_state_dict, build_modules
    from tests.tests_train_util import oracle_key, saved_relu_masks
    from src.util.depth2normal import depth2normal
    monkeypatch.setenv("DINER_TRAIN_BATCH", "1" if batched else "0")
    W = H = 64
    SB, side, K, G, n_cand = 2, 32, 40, 15, 1000
    NR = side * side
    scs = [make_scene(W, H, seed=11 + s) for s in range(SB)]
    msd = make_mlp_state_dict()
    nerf, R = build_modules(scs, msd, torch.device("cuda", 0))
    nerf.train()
    nerf.encoder.latent = nerf.encoder.latent.detach().requires_grad_(True)
    E = torch.stack([s["target_extrinsics"] for s in scs])
    Km = torch.stack([s["target_intrinsics"] for s in scs])
    rays_all = ops.gen_rays(E, Km, W, H, scs[0]["znear"], scs[0]["zfar"], "cuda:0")
    ys, xs = torch.meshgrid(torch.arange(side) + (H - side) // 2, torch.arange(side) + (W - side) // 2, indexing="ij")
    idx = (ys * W + xs).reshape(-1).cuda()
    r = rays_all[:, idx].contiguous()                                       # (SB, NR, 8)
    gen = torch.Generator().manual_seed(77)
    inj = (torch.rand(SB, NR, n_cand, generator=gen).cuda(), torch.randn(SB, NR, G, generator=gen).cuda(),
           torch.rand(SB, NR, K, generator=gen).cuda())
    ren = R(n_samples=K, n_depth_candidates=n_cand, n_gaussian=G, white_bkgd=True)
    with noise.inject(*inj):
        with torch.no_grad():
            z = ren.fill_up_uniform_samples(ren.sample_depthguided(r, nerf, K, n_cand, n_gaussian=G), r)
            ref_out = ren.forward(nerf, r).fine.rgb
        out = ren.forward(nerf, r)
    assert out.fine.rgb.shape == (SB, NR, 3) and out.fine.rgb.requires_grad
    assert max_norm_rel(out.fine.rgb.detach().cpu(), ref_out.cpu()) < 2e-5
    nodes = _field_nodes(out.fine.rgb)
    assert len(nodes) == (1 if batched else SB)
    if len(nodes) == 1 and "Batch" in type(nodes[0]).__name__:      # ABI v6: ONE node for the SB objects, rows object-major in one workspace
        masks = [saved_relu_masks(type("o", (), {"grad_fn": nodes[0]})(), NR * K, obj=sb, n_obj=SB) for sb in range(SB)]
    else:                                                           # DINER_TRAIN_BATCH=0: one node per object
        assert len(nodes) == SB
        masks = [saved_relu_masks(type("o", (), {"grad_fn": n})(), NR * K) for n in nodes]
    Gm = torch.randn(SB, NR, 3, generator=gen)
    (out.fine.rgb * Gm.cuda()).sum().backward()
    assert nerf.encoder.latent.grad.shape == nerf.encoder.latent.shape
    if not batched:
        # one call pair per object (round 5's path, still what mismatched scenes and DINER_TRAIN_BATCH=0 take): the same step as one node must
        # give the same colours bit for bit and the same gradients to summation order -- the oracle comparison below runs on the batched case
        g_obj = {k: p.grad.clone() for k, p in nerf.mlp_fine.named_parameters()}
        l_obj, rgb_obj = nerf.encoder.latent.grad.clone(), out.fine.rgb.detach().clone()
        for p in nerf.mlp_fine.parameters():
            p.grad = None
        nerf.encoder.latent.grad = None
        monkeypatch.setenv("DINER_TRAIN_BATCH", "1")
        with noise.inject(*inj):
            out_b = ren.forward(nerf, r)
        assert len(_field_nodes(out_b.fine.rgb)) == 1 and torch.equal(out_b.fine.rgb.detach(), rgb_obj)
        (out_b.fine.rgb * Gm.cuda()).sum().backward()
        worst = max(max_norm_rel(p.grad.cpu(), g_obj[k].cpu()) for k, p in nerf.mlp_fine.named_parameters())
        e_l = max_norm_rel(nerf.encoder.latent.grad.cpu(), l_obj.cpu())
        print(f"SB=2: one node per object against one node for the objects: parameter gradients {worst:.2e}, d latent {e_l:.2e}, rgb bit-equal")
        assert worst < 1e-5 and e_l < 1e-5
        return

    def oracle(cond):
        w = O.MLPWeights.from_state_dict(msd)
        leaves = {}
        for k, v in vars(w).items():
            for i, t in enumerate(v if isinstance(v, (list, tuple)) else [v]):
                if torch.is_tensor(t) and t.is_floating_point():
                    leaves[(k, i if isinstance(v, (list, tuple)) else None)] = t.requires_grad_(True)
        dlat, rgbs = [], []
        for sb, sc in enumerate(scs):
            normals = depth2normal(sc["depths"], sc["src_intrinsics"])
            Kin = sc["src_intrinsics"]
            scene = O.Scene(latent=sc["latent"].clone().requires_grad_(True), depths=sc["depths"], depths_std=sc["depths_std"], normals=normals,
                            poses=sc["src_extrinsics"], focal=Kin[:, [0, 1], [0, 1]], c=Kin[:, :2, -1], image_shape=sc["image_shape"],
                            feature_padding=sc["feature_padding"])
            rc, zc = r[sb].cpu(), z[sb].cpu()
            xyz = (rc[:, None, :3] + zc[..., None] * rc[:, None, 3:6]).reshape(-1, 3)
            dirs = rc[:, None, 3:6].expand(-1, K, -1).reshape(-1, 3)
            field = O.pixelnerf_forward(scene, w, xyz, dirs, masks[sb] if cond else None).view(NR, K, 4)
            _, rgb_o, _ = O.composite_from_field(field, rc, zc, True)
            (rgb_o * Gm[sb]).sum().backward()                               # weight gradients accumulate over the objects
            dlat.append(scene.latent.grad)
            rgbs.append(rgb_o.detach())
        return leaves, dlat, rgbs

    rms = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm().clamp(min=1e-30))
    for cond in (False, True):
        leaves, dlat, rgbs = oracle(cond)
        for sb in range(SB):
            assert max_norm_rel(out.fine.rgb[sb].detach().cpu(), rgbs[sb]) < 2e-5
        worst = max(((n, max_norm_rel(p.grad.cpu(), leaves[oracle_key(n)].grad), rms(p.grad.cpu(), leaves[oracle_key(n)].grad))
                     for n, p in nerf.mlp_fine.named_parameters()), key=lambda t: t[1])
        e_lat = max(max_norm_rel(nerf.encoder.latent.grad[sb].cpu(), dlat[sb]) for sb in range(SB))
        r_lat = max(rms(nerf.encoder.latent.grad[sb].cpu(), dlat[sb]) for sb in range(SB))
        r_par = max(rms(p.grad.cpu(), leaves[oracle_key(n)].grad) for n, p in nerf.mlp_fine.named_parameters())
        print(f"SB=2 x {side}x{side} patch, {'conditioned on the HIP relu decisions' if cond else 'unconditioned'}: worst parameter gradient "
              f"{worst[0]} max-norm {worst[1]:.2e} (Frobenius {r_par:.2e}), d latent per object max-norm {e_lat:.2e} (Frobenius {r_lat:.2e})")
        if cond:
            assert worst[1] < TOL_GRAD and e_lat < TOL_GRAD
        else:
            assert r_par < 3e-3 and r_lat < 3e-3
    # the per-object slabs of the latent gradient are different (each object's own rays and maps)
    assert not torch.equal(nerf.encoder.latent.grad[0], nerf.encoder.latent.grad[1])


def test_shipped_ray_batch_equals_the_sum_of_reference_sized_batches(ops, monkeypatch):
    """The ray batch the shipped configs train on (4096 rays per object = a 64 x 64 patch, diner.py:57 with configs/train_dtu.yaml:63; 163,840
    sample points, 655,360 per-view rows) takes launch plans no oracle comparison reaches -- 128-row tiles of the forward / data-gradient
    products, 256 x 256 weight-gradient tiles over 64 row chunks, the two-segment fc_1 + lin_z product -- and a CPU autograd of that size needs
    40 GB.  Size-independent property instead: rays are independent and the loss is a sum over rays, so the gradients of the 4096-ray step
    are the SUM of the gradients of its 32 sub-batches of 128 rays -- the size whose gradients are pinned against the oracle's autograd at 1e-4
    (test_field_forward_and_backward_against_oracle_autograd[5120]).  Per row the arithmetic is the same whatever the tile shape, so the relu
    decisions are the same and only the summation order of the weight gradients differs: 1e-4 max-norm per tensor, rgb bit for bit.
    (Round 5: "the same arithmetic per row" holds within ONE forward -- by its size rule the host would take the fused forward for the
    4096-ray batch and the layer-wise one for a 128-ray batch on this scene; the fused forward, what the shipped step runs, is forced for both.
    The layer-wise forward against the same sums: DINER_TRAIN_FUSED_FWD=0 in the environment of the whole file, tools/README.md.)"""
    from tests.test_boundary_gpu import setup_model
    from diner_amd import noise
    if os.environ.get("DINER_TRAIN_FUSED_FWD", "") != "0":
        monkeypatch.setenv("DINER_TRAIN_FUSED_FWD", "1")
    sc, nerf, R, rays = setup_model(64, 64, 8)
    nerf.train()
    NR, K, G, n_cand, CH = 4096, 40, 15, 1000, 128
    r = rays.cuda()[None]                                                   # all 64 x 64 rays of the target view
    assert r.shape[1] == NR
    gen = torch.Generator().manual_seed(5)
    inj = (torch.rand(1, NR, n_cand, generator=gen).cuda(), torch.randn(1, NR, G, generator=gen).cuda(), torch.rand(1, NR, K, generator=gen).cuda())
    Gm = torch.randn(1, NR, 3, generator=gen).cuda()
    ren = R(n_samples=K, n_depth_candidates=n_cand, n_gaussian=G, white_bkgd=True)
    nerf.encoder.latent = nerf.encoder.latent.detach().requires_grad_(True)
    params = dict(nerf.mlp_fine.named_parameters())

    def step(lo, hi):
        for p in params.values():
            p.grad = None
        nerf.encoder.latent.grad = None
        with noise.inject(*(t[:, lo:hi] for t in inj)):
            out = ren.forward(nerf, r[:, lo:hi])
        (out.fine.rgb * Gm[:, lo:hi]).sum().backward()
        return out.fine.rgb.detach(), {k: p.grad.clone() for k, p in params.items()}, nerf.encoder.latent.grad.clone()

    rgb_full, g_full, l_full = step(0, NR)
    assert torch.isfinite(rgb_full).all() and all(torch.isfinite(v).all() for v in g_full.values())
    g_sum = {k: torch.zeros_like(v) for k, v in g_full.items()}
    l_sum = torch.zeros_like(l_full)
    for lo in range(0, NR, CH):
        rgb_c, g_c, l_c = step(lo, lo + CH)
        assert torch.equal(rgb_c, rgb_full[:, lo:lo + CH]), f"rays {lo}..{lo + CH}: the forward depends on the batch the ray is rendered in"
        for k in g_sum:
            g_sum[k] += g_c[k]
        l_sum += l_c
    worst = max(((k, max_norm_rel(g_full[k].cpu(), g_sum[k].cpu())) for k in g_sum), key=lambda t: t[1])
    e_lat = max_norm_rel(l_full.cpu(), l_sum.cpu())
    print(f"4096-ray step against the sum of its 32 sub-batches of 128 rays: worst parameter gradient {worst[0]} {worst[1]:.2e}, d latent {e_lat:.2e}; rgb bit-equal")
    assert worst[1] < TOL_GRAD and e_lat < TOL_GRAD


def test_shipped_step_four_objects_equals_the_sum_of_its_objects(ops):
    """The whole step the shipped configs run (configs/train_dtu.yaml:16,63: SB = 4 objects x 4096 rays x 40 samples; DINER.calc_losses,
    diner.py:217-290: ONE renderer.forward on (4, 4096, 8) rays): 60 GB of saved activations alive between forward and backward, the SB
    per-object field nodes, one autograd node for the objects' latent slabs (diner_amd.train._ObjectSlabs), autograd's sum of the four
    gradient sets of the shared MLP parameters.  Property: objects are independent, so the step's parameter gradients are the sum of the four
    single-object steps' and its latent gradient is their stack (the single-object 4096-ray step is tied to the oracle by
    test_shipped_ray_batch_equals_the_sum_of_reference_sized_batches)."""
    from diner_amd import noise
    from diner_amd.synthetic import make_scene, make_mlp_state_dict, build_modules
    W = H = 64
    SB, NR, K, G, n_cand = 4, 4096, 40, 15, 1000
    scs = [make_scene(W, H, seed=21 + s) for s in range(SB)]
    msd = make_mlp_state_dict()
    dev = torch.device("cuda", 0)
    E = torch.stack([s["target_extrinsics"] for s in scs])
    Km = torch.stack([s["target_intrinsics"] for s in scs])
    rays = ops.gen_rays(E, Km, W, H, scs[0]["znear"], scs[0]["zfar"], dev)                  # (SB, 4096, 8)
    gen = torch.Generator().manual_seed(13)
    inj = (torch.rand(SB, NR, n_cand, generator=gen).cuda(), torch.randn(SB, NR, G, generator=gen).cuda(), torch.rand(SB, NR, K, generator=gen).cuda())
    Gm = torch.randn(SB, NR, 3, generator=gen).cuda()

    def run(objs):
        nerf, R = build_modules([scs[i] for i in objs], msd, dev)
        nerf.train()
        nerf.encoder.latent = nerf.encoder.latent.detach().requires_grad_(True)
        ren = R(n_samples=K, n_depth_candidates=n_cand, n_gaussian=G, white_bkgd=True)
        idx = torch.tensor(objs, device=dev)
        with noise.inject(*(t[idx] for t in inj)):
            out = ren.forward(nerf, rays[idx])
        (out.fine.rgb * Gm[idx]).sum().backward()
        g = {k: p.grad.clone() for k, p in nerf.mlp_fine.named_parameters()}
        return out.fine.rgb.detach().clone(), g, nerf.encoder.latent.grad.clone()

    torch.cuda.reset_peak_memory_stats()
    rgb_all, g_all, l_all = run(list(range(SB)))
    peak = torch.cuda.max_memory_allocated() / 2 ** 30
    assert rgb_all.shape == (SB, NR, 3) and l_all.shape[0] == SB and torch.isfinite(rgb_all).all()
    g_sum = {k: torch.zeros_like(v) for k, v in g_all.items()}
    for sb in range(SB):
        rgb1, g1, l1 = run([sb])
        assert torch.equal(rgb1[0], rgb_all[sb])
        # (the latent gradient is scattered with float atomics: the order of the additions is not reproducible, the values are to round-off)
        assert max_norm_rel(l_all[sb].cpu(), l1[0].cpu()) < 1e-5, f"object {sb}: latent gradient slab differs from the single-object step's"
        for k in g_sum:
            g_sum[k] += g1[k]
    worst = max(((k, max_norm_rel(g_all[k].cpu(), g_sum[k].cpu())) for k in g_sum), key=lambda t: t[1])
    print(f"SB = 4 x 4096 rays x 40 samples: peak device memory {peak:.1f} GiB; parameter gradients against the sum of the four single-object steps: "
          f"worst {worst[0]} {worst[1]:.2e}; latent gradient slabs within 1e-5, rgb bit-equal per object")
    assert worst[1] < 1e-5


def test_touched_texel_projection_equals_the_whole_map_projection(ops, monkeypatch):
    """Round 6: the fused training forward projects through lin_z[0..2] only the latent rows its batch touches (k_mark_rows / k_compact_rows,
    a product over a device-side row count, rows scattered back).  Per row the product is the same arithmetic as the whole-map projection, so
    outputs must be BIT-EQUAL (and the gradients equal to the round-off of their atomics) -- with the list (default), with the whole map (DINER_TRAIN_PROJ_TOUCHED=0) and when the
    list overflows its capacity and the dense projection takes over ON THE DEVICE (DINER_TRAIN_PROJ_CAP=64: the flag `dense`, no host decision);
    the latent gradient (float atomics in the scatter) to round-off."""
    from diner_amd import train
    from diner_amd.synthetic import make_scene, make_mlp_state_dict
    from tests.tests_train_util import module_param_list
    from src.util.depth2normal import depth2normal
    monkeypatch.setenv("DINER_TRAIN_FUSED_FWD", "1")
    sc = make_scene(64, 64, seed=3)
    sc["normals"] = depth2normal(sc["depths"], sc["src_intrinsics"])
    Kin = sc["src_intrinsics"]
    P = 5120
    g = torch.Generator().manual_seed(11)
    xyz = (torch.rand(P, 3, generator=g) - 0.5) * 0.2          # a compact cloud at the object: a few hundred texels per view, far below the list's capacity
    dirs = torch.nn.functional.normalize(torch.randn(P, 3, generator=g), dim=-1)
    Gm = torch.randn(P, 4, generator=g).cuda()
    msd = make_mlp_state_dict()
    res = {}
    for tag, env in (("touched", {}), ("whole", {"DINER_TRAIN_PROJ_TOUCHED": "0"}), ("overflow", {"DINER_TRAIN_PROJ_CAP": "64"})):
        for k in ("DINER_TRAIN_PROJ_TOUCHED", "DINER_TRAIN_PROJ_CAP"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        train.release_buffers()
        params, _ = module_param_list(msd)
        lat = sc["latent"].cuda().requires_grad_(True)
        scene = ops.HipScene(lat.detach(), sc["depths"].cuda(), sc["depths_std"].cuda(), sc["normals"].cuda(), sc["src_extrinsics"],
                             Kin[:, [0, 1], [0, 1]], Kin[:, :2, -1], sc["image_shape"], sc["feature_padding"])
        out = train.field_train(scene, xyz.cuda(), dirs.cuda(), lat, params)
        (out * Gm).sum().backward()
        res[tag] = (out.detach().clone(), [p.grad.clone() for p in params], lat.grad.clone())
    o0, g0, l0 = res["touched"]
    assert torch.isfinite(o0).all() and all(torch.isfinite(t).all() for t in g0)
    for tag in ("whole", "overflow"):
        o, gs, l = res[tag]
        print(f"{tag}: forward max |d| {float((o - o0).abs().max()):.3e} ({int((o != o0).any(-1).sum())} of {P} points differ), worst parameter gradient "
              f"{max(max_norm_rel(a.cpu(), b.cpu()) for a, b in zip(gs, g0)):.3e}, d latent {max_norm_rel(l.cpu(), l0.cpu()):.3e}")
        assert torch.equal(o, o0), f"{tag}: the forward differs from the touched-rows projection"
        # (the backward is the same launch sequence on bit-equal saved activations; lin_in's / lin_out's weight gradients and the latent scatter
        # sum with float atomics, so their last bits are not reproducible from run to run)
        assert max(max_norm_rel(a.cpu(), b.cpu()) for a, b in zip(gs, g0)) < 1e-5, f"{tag}: parameter gradients differ"
        assert max_norm_rel(l.cpu(), l0.cpu()) < 1e-5


def test_backward_routes_of_round_6_agree(ops, monkeypatch):
    """Round 6 gave the batched backward three new routes, each behind a switch read per call: the lin_z adjoint in map space over the touched
    texel rows (DINER_TRAIN_LINZ_MAPSPACE), its scatter over columns sorted by texel (DINER_TRAIN_SCATTER_SORTED) and block 2's fc_1 data
    gradient once per point behind the view mean + its weight gradient over view-mean activations (DINER_TRAIN_VIEW_SHARED).  With a switch at 0 the step runs the round-5 launch sequence for
    that part; forward outputs are bit-equal (the forward does not change), all gradients agree to round-off (float atomics, another
    summation order; the map-space dWz is a bf16x6 product over texel rows where the sample-space one is f16x3 over sample rows)."""
    from diner_amd import train
    from diner_amd.synthetic import make_scene, make_mlp_state_dict
    from tests.tests_train_util import module_param_list
    from src.util.depth2normal import depth2normal
    monkeypatch.setenv("DINER_TRAIN_FUSED_FWD", "1")
    sc = make_scene(64, 64, seed=5)
    sc["normals"] = depth2normal(sc["depths"], sc["src_intrinsics"])
    Kin = sc["src_intrinsics"]
    P = 5120                                                   # 20480 per-view rows: the map-space route takes batches of >= 4096
    g = torch.Generator().manual_seed(13)
    xyz = (torch.rand(P, 3, generator=g) - 0.5) * 0.2
    dirs = torch.nn.functional.normalize(torch.randn(P, 3, generator=g), dim=-1)
    Gm = torch.randn(1, P, 4, generator=g).cuda()
    msd = make_mlp_state_dict()
    names = ("DINER_TRAIN_LINZ_MAPSPACE", "DINER_TRAIN_SCATTER_SORTED", "DINER_TRAIN_VIEW_SHARED")
    switches = tuple(n + "=0" for n in names) + ("DINER_TRAIN_VIEW_SHARED=2",)      # = 2: the data gradient once per point, the weight gradient over all rows
    res = {}
    for tag in ("default",) + switches:
        for k in names:
            monkeypatch.delenv(k, raising=False)
        if tag != "default":
            monkeypatch.setenv(*tag.split("="))
        train.release_buffers()
        params, _ = module_param_list(msd)
        lat = sc["latent"].cuda().requires_grad_(True)
        scene = ops.HipScene(lat.detach(), sc["depths"].cuda(), sc["depths_std"].cuda(), sc["normals"].cuda(), sc["src_extrinsics"],
                             Kin[:, [0, 1], [0, 1]], Kin[:, :2, -1], sc["image_shape"], sc["feature_padding"])
        out = train.field_train_batch([scene], xyz.cuda()[None], dirs.cuda()[None], lat[None], params)
        (out * Gm).sum().backward()
        res[tag] = (out.detach().clone(), [p.grad.clone() for p in params], lat.grad.clone())
    o0, g0, l0 = res["default"]
    assert torch.isfinite(o0).all() and all(torch.isfinite(t).all() for t in g0) and torch.isfinite(l0).all()
    for tag in switches:
        o, gs, l = res[tag]
        worst = max(max_norm_rel(a.cpu(), b.cpu()) for a, b in zip(gs, g0))
        print(f"{tag}: worst parameter gradient {worst:.3e}, d latent {max_norm_rel(l.cpu(), l0.cpu()):.3e}")
        assert torch.equal(o, o0), f"{tag}: the forward differs"
        assert worst < 1e-5, f"{tag}: parameter gradients differ"      # measured 6.6e-7 - 7.6e-7
        assert max_norm_rel(l.cpu(), l0.cpu()) < 1e-5, f"{tag}: latent gradient differs"      # measured 3.3e-7 - 8.9e-7


def test_channels_last_latent_is_taken_and_returned_without_a_copy(ops, monkeypatch):
    """Round 6: a latent that already lies channels-last in memory (NCHW shape, channels-last strides -- what image_encoder.py emits on a HIP
    device) is read in place by HipScene and its gradient comes back as a view of the backward's channels-last buffer; an NCHW-contiguous latent
    keeps the two transposing copies.  Same forward bits, same gradients (to the round-off of the scatter's atomics), on the per-object node and
    on the batched node."""
    from diner_amd import train
    from diner_amd.synthetic import make_scene, make_mlp_state_dict, as_encoded
    from tests.tests_train_util import module_param_list
    from src.util.depth2normal import depth2normal
    sc = make_scene(64, 64, seed=7)
    sc["normals"] = depth2normal(sc["depths"], sc["src_intrinsics"])
    Kin = sc["src_intrinsics"]
    g = torch.Generator().manual_seed(17)
    msd = make_mlp_state_dict()
    for batched, P in ((False, 300), (True, 5120)):
        if batched:
            monkeypatch.setenv("DINER_TRAIN_FUSED_FWD", "1")
        xyz = ((torch.rand(P, 3, generator=g) - 0.5) * 0.2).cuda()
        dirs = torch.nn.functional.normalize(torch.randn(P, 3, generator=g), dim=-1).cuda()
        Gm = torch.randn(P, 4, generator=g).cuda()
        res = {}
        for fmt in ("nchw", "channels_last"):
            train.release_buffers()
            params, _ = module_param_list(msd)
            lat = sc["latent"].cuda()
            if fmt == "channels_last":
                lat = as_encoded(lat[None])[0]
            lat.requires_grad_(True)
            scene = ops.HipScene(lat.detach(), sc["depths"].cuda(), sc["depths_std"].cuda(), sc["normals"].cuda(), sc["src_extrinsics"],
                                 Kin[:, [0, 1], [0, 1]], Kin[:, :2, -1], sc["image_shape"], sc["feature_padding"])
            assert (scene.latent_cl.data_ptr() == lat.data_ptr()) == (fmt == "channels_last")
            out = train.field_train_batch([scene], xyz[None], dirs[None], lat[None], params)[0] if batched else train.field_train(scene, xyz, dirs, lat, params)
            (out * Gm).sum().backward()
            assert train._channels_last(lat.grad) if fmt == "channels_last" else lat.grad.is_contiguous()
            res[fmt] = (out.detach().clone(), lat.grad.clone(), [p.grad.clone() for p in params])
        (o0, l0, g0), (o1, l1, g1) = res["nchw"], res["channels_last"]
        assert torch.equal(o0, o1)
        assert max_norm_rel(l1.cpu(), l0.cpu()) < 1e-5
        assert max(max_norm_rel(a.cpu(), b.cpu()) for a, b in zip(g1, g0)) < 1e-5
