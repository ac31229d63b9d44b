import sys, torch
sys.path.insert(0, "/root/repo")
from tests.test_hip_parity import *          # noqa
from diner_amd import ops
g = load("g6_pixelnerf.npz")
sc, scene, w, msd, rays = oracle_setup(int(g["W"]), int(g["H"]), int(g["seed"]))
hs = hip_scene(ops, sc)
pts, dirs = T(g["pts"]).cuda(), T(g["dirs"]).cuda()
MAG = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0e5
for key in ("blocks.1.fc_0.bias", "blocks.3.fc_0.bias", "blocks.4.fc_0.bias", "lin_in.bias"):
    big = {k: v.clone() for k, v in msd.items()}
    big[key] = big[key] + MAG * (torch.arange(512) % 7 == 0)
    hm = hip_mlp(ops, big)
    exact = ops.field_from_points(hs, hm, pts, dirs, precision="fp32")
    for mode in ("f16x3", "f16"):
        got = ops.field_from_points(hs, hm, pts, dirs, precision=mode)
        print(key, mode, "finite", bool(torch.isfinite(got).all()), "equal", bool(torch.equal(got, exact)),
              "max diff", float((got - exact).abs().max()), "fallbacks", hm.fallback_launches())
