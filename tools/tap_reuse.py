"""How many distinct feature-map texels do the 16 consecutive samples of a tile touch per source view?  (CPU, oracle sampler.)
Decides whether de-duplicating taps per tile can take the lin_z gather off the vector-memory path (profiles/r02_kernel_experiments.md)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import diner_oracle as O
from tests.helpers import oracle_setup

W, H, K, NC = (int(sys.argv[1]), int(sys.argv[2])) + (128, 1000) if len(sys.argv) > 2 else (800, 600, 128, 1000)
G = int(15 * K / 40)
sc, scene, w, msd, rays = oracle_setup(W, H, 0)
g = torch.Generator().manual_seed(1)
idx = torch.randperm(rays.shape[0], generator=g)[:256]
r = rays[idx]
z = O.sample_depthguided(scene, r, K, NC, G, torch.rand(len(r), NC, generator=g), torch.randn(len(r), G, generator=g))
z = O.fill_up_uniform_samples(z, r, torch.rand(len(r), K, generator=g)) if hasattr(O, "fill_up_uniform_samples") else z
z = z.sort(-1).values
pts = r[:, None, :3] + z[..., None] * r[:, None, 3:6]                      # (R, K, 3)
xc = O.world_to_cam(scene, pts.reshape(-1, 3))                              # (NV, R*K, 3)
uv = O.project_uv(scene, xc)                                                # (NV, N, 2) in [-1, 1]
Hf, Wf = scene.latent.shape[-2:]
size = torch.tensor([Wf, Hf], dtype=torch.float32)
uv = uv * ((size - scene.feature_padding * 2) / size).view(1, 1, 2)
px = ((uv + 1) * size.view(1, 1, 2) - 1) / 2                                # align_corners=False pixel coordinates
x0 = px.floor().long()
NV = px.shape[0]
uniq = []
for v in range(NV):
    b = x0[v].reshape(len(r), K // 16, 16, 2)
    for dx in (0, 1):
        pass
    taps = torch.stack([(b[..., 1] + dy).clamp(0, Hf - 1) * Wf + (b[..., 0] + dx).clamp(0, Wf - 1) for dy in (0, 1) for dx in (0, 1)], -1)
    taps = taps.reshape(len(r), K // 16, 64).numpy()
    uniq += [len(np.unique(t)) for rr in taps for t in rr]
u = np.array(uniq)
print(f"{W}x{H} K={K}: feature map {Wf}x{Hf}; unique texels per (tile of 16 samples, view): mean {u.mean():.1f}  median {np.median(u):.0f}  "
      f"p90 {np.percentile(u, 90):.0f}  max {u.max()}  share <=16: {(u <= 16).mean():.3f}  <=32: {(u <= 32).mean():.3f}")
print("histogram (bins of 8):", np.histogram(u, bins=[0, 8, 16, 24, 32, 40, 48, 56, 65])[0].tolist())
per_tile = u.reshape(NV, -1).max(0)          # a workgroup's tile = the same 16 samples in all views
print(f"max over the {NV} views of a tile: share <=16: {(per_tile <= 16).mean():.3f}  <=32: {(per_tile <= 32).mean():.3f}  mean {per_tile.mean():.1f}")
