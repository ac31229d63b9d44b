"""Dump the per-view kernel's hand-over buffer (mean hidden state per point, 512 floats) for one fixed input.
usage: DINER_AMD_LIB=... DINER_AMD_PRECISION=f16x3n python tools/diag_h3n_xpre.py TAG  -> gpurun_out/xpre_TAG.npy"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from diner_amd import ops
import test_hip_parity as T

g = T.load("g6_pixelnerf.npz")
sc, scene, w, msd, rays = T.oracle_setup(int(g["W"]), int(g["H"]), int(g["seed"]))
hs, hm = T.hip_scene(ops, sc), T.hip_mlp(ops, msd)
keep = []
orig = ops._workspace
def grab(n, dev):
    t = orig(n, dev); t.zero_(); keep.append(t); return t
ops._workspace = grab
out = ops.field_from_points(hs, hm, T.T(g["pts"]).cuda(), T.T(g["dirs"]).cuda())
torch.cuda.synchronize()
P = out.shape[0]
x = keep[-1].view(torch.float32)[: P * 512].cpu().numpy().reshape(P // 16, 32, 64, 4)
os.makedirs("gpurun_out", exist_ok=True)
np.save(f"gpurun_out/xpre_{sys.argv[1]}.npy", x)
print(sys.argv[1], "field err", T.max_norm_rel(out.cpu(), g["out"]))
