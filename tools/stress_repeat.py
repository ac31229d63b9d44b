"""Stress: the same field evaluation (12 replicas of the 512 fixture points, several tiles per workgroup) repeated many
times in every arithmetic mode; every run and every replica must be bit-identical to the first."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from diner_amd import ops
import tests.test_hip_parity as T
g = T.load("g6_pixelnerf.npz")
sc, scene, w, msd, rays = T.oracle_setup(int(g["W"]), int(g["H"]), int(g["seed"]))
hs, hm = T.hip_scene(ops, sc), T.hip_mlp(ops, msd)
R = 12
pts, dirs = T.T(g["pts"]).repeat(R, 1).cuda(), T.T(g["dirs"]).repeat(R, 1).cuda()
big_p = T.T(g["pts"]).repeat(400, 1).cuda(); big_d = T.T(g["dirs"]).repeat(400, 1).cuda()     # 204800 points: 50 tiles per CU
for mode, name in (("f16x3", "f16x3"), ("fp32", "fp32"), ("f16", "f16")):
    ops.set_precision(mode)
    first = ops.field_from_points(hs, hm, pts, dirs).view(R, -1, 4)
    bad = 0
    for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
        o = ops.field_from_points(hs, hm, pts, dirs).view(R, -1, 4)
        bad += int(not all(torch.equal(o[r], first[0]) for r in range(R)))
    ob = ops.field_from_points(hs, hm, big_p, big_d).view(400, -1, 4)
    bad_big = int((ob != first[0][None]).any(dim=(1, 2)).sum())
    print(f"{name}: {bad} differing runs of the small case, {bad_big} of 400 replicas differ in the 204800-point case")
