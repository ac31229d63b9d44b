"""Timing of one training step of the HIP training path (row f1): renderer.forward + backward of a rgb loss for
ray_batch_size rays x 40 samples x 4 views (configs/train_dtu.yaml:55-65), MLP parameters and encoder.latent gradients."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from tests.test_boundary_gpu import setup_model
from diner_amd import ops
sc, nerf, R, rays = setup_model(64, 64, 0)
nerf.train()
nerf.encoder.latent = nerf.encoder.latent.detach().requires_grad_(True)
SIZES = [int(a) for a in sys.argv[1:]] or [128, 512, 2048]
for NR in SIZES:
    K, G = 40, 15
    r = rays[torch.linspace(0, rays.shape[0] - 1, NR).long()].cuda()[None]
    ren = R(n_samples=K, n_depth_candidates=1000, n_gaussian=G, white_bkgd=True)
    def step():
        for p in nerf.parameters(): p.grad = None
        nerf.encoder.latent.grad = None
        out = ren.forward(nerf, r)
        out.fine.rgb.square().mean().backward()
    step(); torch.cuda.synchronize()
    t = time.perf_counter()
    n = 10
    for _ in range(n): step()
    host = (time.perf_counter() - t) / n                   # host time to enqueue a step (the device is still working)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / n
    flop = 3 * 2 * NR * K * (4 * (55 * 512 + 9 * 512 * 512) + 4 * 512 * 512 + 4 * 512)      # fwd + dgrad + wgrad
    print(f"{NR} rays x {K} samples: {dt*1e3:.2f} ms per forward+backward step = {NR/dt:.0f} rays/s, {flop/dt/1e12:.1f} TFLOP/s fp32; host enqueue {host*1e3:.2f} ms per step")
