"""Time k_hoist_linz alone (small code footprint: one 32-stage layer body in a runtime loop) to compare its MFMA
efficiency with the big field kernels."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diner_amd import ops
from diner_amd.synthetic import make_scene, make_mlp_state_dict
from src.util.depth2normal import depth2normal
dev = torch.device("cuda", 0)
W, H = 400, 300
sc = make_scene(W, H, seed=0)
normals = depth2normal(sc["depths"], sc["src_intrinsics"])
Kin = sc["src_intrinsics"]
scene = ops.HipScene(sc["latent"].to(dev), sc["depths"].to(dev), sc["depths_std"].to(dev), normals.to(dev),
                     sc["src_extrinsics"], Kin[:, [0, 1], [0, 1]], Kin[:, :2, -1], sc["image_shape"], sc["feature_padding"])
mlp = ops.HipMlp({k: v.to(dev) for k, v in make_mlp_state_dict().items()})
for _ in range(3):
    scene.prepare(mlp, force=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
N = 10
for _ in range(N):
    scene.prepare(mlp, force=True)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / N
rows = scene.nv * scene.Hf * scene.Wf
flop = rows * ops.FLOP_HOIST_PER_PIXEL
print(f"k_hoist_linz: {rows} rows, {ms:.3f} ms, {flop/ms/1e9:.1f} TFLOP/s = {flop/ms/1e9/157.3:.3f} of fp32 MFMA peak")
