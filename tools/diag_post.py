import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from diner_amd import ops
import test_hip_parity as T
g = T.load("g6_pixelnerf.npz")
sc, scene, w, msd, rays = T.oracle_setup(int(g["W"]), int(g["H"]), int(g["seed"]))
hs, hm = T.hip_scene(ops, sc), T.hip_mlp(ops, msd)
out = ops.field_from_points(hs, hm, T.T(g["pts"]).cuda(), T.T(g["dirs"]).cuda()).cpu()
ref = torch.as_tensor(g["out"]).float()
bad = ~torch.isfinite(out)
print("nan count per channel", bad.sum(0).tolist(), "of", out.shape[0])
print("first rows out:", out[:3].tolist()); print("ref:", ref[:3].tolist())
d = (out - ref).abs(); d[bad] = 0
print("max abs diff per channel (finite)", d.amax(0).tolist())
idx = bad.any(1).nonzero().flatten()[:20].tolist(); print("nan points", idx)
