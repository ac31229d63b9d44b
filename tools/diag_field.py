"""Diagnostic (GPU box): per-sample field error on the e2e fixture with the reference's z."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from oracle import diner_oracle as O
from tests.helpers import load, oracle_setup
from tests.test_hip_parity import hip_scene, hip_mlp, T
from diner_amd import ops

g = load("g8_render_cfg1.npz")
W, H, K = int(g["W"]), int(g["H"]), int(g["K"])
sc, scene, w, msd, rays = oracle_setup(W, H, int(g["seed"]))
hs, hm = hip_scene(ops, sc), hip_mlp(ops, msd)
ref_z = T(g["z"]); rc = rays.cuda()
field = ops.field_from_rays(hs, hm, rc, ref_z.cuda()).cpu()
wts, rgb, depth = ops.composite(field.cuda(), ref_z.cuda(), rc, False)
e = (rgb.cpu() - T(g["rgb"])).abs().max(-1).values
worst = e.argsort(descending=True)[:6]
print("worst rays", worst.tolist(), e[worst].tolist())
for r in worst[:3].tolist():
    pts = rays[r, None, :3] + ref_z[r].unsqueeze(-1) * rays[r, None, 3:6]
    dirs = rays[r, None, 3:6].expand(K, -1)
    fo = O.pixelnerf_forward(scene, w, pts, dirs)
    d = (field[r] - fo).abs()
    print(f"ray {r}: field err max per channel {d.max(0).values.tolist()}, at samples {d.argmax(0).tolist()}")
    k = int(d[:, 3].argmax())
    print(f"   sample {k}: hip {field[r,k].tolist()} oracle {fo[k].tolist()} z={ref_z[r,k].item()}")
    # raw features
    zx = O.mlp_input(scene, pts[k:k+1], dirs[k:k+1])    # (4,1,567)
    raw_hip = ops.mlp_forward(hm, zx.cuda()).cpu()
    raw_or = O.mlp_forward(w, zx)
    print(f"   mlp on oracle features: hip {raw_hip.tolist()} oracle {raw_or.tolist()}")
    # double-precision oracle of the MLP to see who is closer
    import copy
    wd = copy.deepcopy(w)
    for name in ("lin_in_w","lin_in_b","lin_out_w","lin_out_b"):
        setattr(wd, name, getattr(wd, name).double())
    for name in ("lin_z_w","lin_z_b","fc0_w","fc0_b","fc1_w","fc1_b"):
        setattr(wd, name, [t.double() for t in getattr(wd, name)])
    print(f"   fp64 mlp on the same features: {O.mlp_forward(wd, zx.double()).tolist()}")

print("---- composite isolation on ray 1207")
r = 1207
pts = rays[r, None, :3] + ref_z[r].unsqueeze(-1) * rays[r, None, 3:6]
dirs = rays[r, None, 3:6].expand(K, -1)
fo = O.pixelnerf_forward(scene, w, pts, dirs)
w_a, rgb_a, d_a = O.composite_from_field(field[r:r+1], rays[r:r+1], ref_z[r:r+1], False)   # HIP field, oracle composite
w_b, rgb_b, d_b = O.composite_from_field(fo[None], rays[r:r+1], ref_z[r:r+1], False)       # oracle field, oracle composite
w_c, rgb_c, d_c = ops.composite(fo[None].cuda(), ref_z[r:r+1].cuda(), rays[r:r+1].cuda(), False)  # oracle field, HIP composite
print("golden rgb", g["rgb"][r], "depth", g["depth"][r])
print("HIP field + oracle comp", rgb_a.tolist(), d_a.tolist())
print("orc field + oracle comp", rgb_b.tolist(), d_b.tolist())
print("orc field + HIP comp   ", rgb_c.cpu().tolist(), d_c.cpu().tolist())
print("z", ref_z[r].tolist())
print("sigma hip", field[r,:,3].tolist())
print("weights oracle", w_b[0].tolist())
