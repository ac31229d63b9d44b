"""Debug aid: the latent map projected by the training forward (f16x3 products of the 512-layer kernel) against diner_scene_prepare_f32."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from diner_amd import ops, train
from diner_amd.synthetic import make_scene, make_mlp_state_dict
from tests.tests_train_util import module_param_list

for (W, H, P) in ((64, 64, 200), (64, 64, 5120), (400, 300, 8192)):
    sc = make_scene(W, H, seed=3)
    msd = make_mlp_state_dict()
    hs = ops.HipScene(sc["latent"].cuda(), sc["depths"].cuda(), sc["depths_std"].cuda(), None, sc["src_extrinsics"],
                      sc["src_intrinsics"][:, [0, 1], [0, 1]], sc["src_intrinsics"][:, :2, -1], sc["image_shape"], sc["feature_padding"])
    params, names = module_param_list(msd)
    mlp = train._step_mlp([p.detach() for p in params], 6.28)
    hs.prepare(mlp, force=True)
    A = hs.latent_proj.clone()
    rows = hs.nv * hs.Hf * hs.Wf
    g = torch.Generator().manual_seed(1)
    xyz = (torch.rand(P, 3, generator=g) - 0.5).cuda()
    dirs = torch.nn.functional.normalize(torch.randn(P, 3, generator=g), dim=-1).cuda()
    os.environ["DINER_TRAIN_FUSED_FWD"] = "1"
    latent = sc["latent"].cuda().requires_grad_(True)
    out = train.field_train(hs, xyz, dirs, latent, params)
    B = next(iter(train._PROJ.values())).view(torch.float32)[:A.numel()]
    d = (A - B).view(3, rows, 512).abs()
    print(f"{W}x{H} rows {rows} P {P}: max |A| {float(A.abs().max()):.3g}, max diff per plane {[float(x) for x in d.amax(dim=(1, 2))]}")
    bad = (d.amax(dim=2) > 1e-3)
    for b in range(3):
        idx = bad[b].nonzero().flatten()
        if len(idx):
            print(f"   plane {b}: {len(idx)} bad rows, first {idx[:5].tolist()} last {idx[-5:].tolist()}")
    os.environ["DINER_TRAIN_FUSED_FWD"] = "0"
    out0 = train.field_train(hs, xyz, dirs, latent, params)
    print("   forward fused vs layer-wise:", float((out - out0).abs().max()))
