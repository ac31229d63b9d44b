"""Replicate the 512 fixture points R times (several tiles per workgroup) and check every replica equals the first."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from diner_amd import ops
import test_hip_parity as T
g = T.load("g6_pixelnerf.npz")
sc, scene, w, msd, rays = T.oracle_setup(int(g["W"]), int(g["H"]), int(g["seed"]))
hs, hm = T.hip_scene(ops, sc), T.hip_mlp(ops, msd)
R = int(sys.argv[1]) if len(sys.argv) > 1 else 32
pts = T.T(g["pts"]).repeat(R, 1).cuda(); dirs = T.T(g["dirs"]).repeat(R, 1).cuda()
out = ops.field_from_points(hs, hm, pts, dirs).cpu().view(R, 512, 4)
ref = torch.as_tensor(g["out"]).float()
err = (out - ref[None]).abs().amax(dim=(1, 2)) / ref.abs().max()
print("replica errors:", " ".join(f"{e:.1e}" for e in err.tolist()))
