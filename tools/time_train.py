"""Timing of one training step of the HIP training path (row f1) at the step the shipped configs run:

    python tools/time_train.py [--objects 4] [--rays 4096] [--samples 40] [--size 400x300] [--steps 5]

reference: DINER.calc_losses (src/models/diner.py:217-290) with configs/train_dtu.yaml:16,52-63 (batch_size 4, w_vgg 0.1 =>
ray_batch_size = vgg_spatch^2 = 4096 rays per object, diner.py:57; n_samples 40, n_gaussian 15, 1000 candidates): SB objects, a
64 x 64 pixel patch of rays per object (diner.py:233-247), ONE renderer.forward on (SB, 4096, 8) rays, MSE loss on fine.rgb,
backward into the MLP parameters and encoder.latent.  Smaller --objects / --rays give the earlier rounds' batches
(128 rays x 1 object is the w_vgg = 0 default of the constructor, not what any shipped config runs).

Prints ms per step, rays/s, TFLOP/s fp32-equivalent (forward + data gradient + weight gradient of the reference's FLOPs),
the saved-activation workspace per object and the peak device memory."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from diner_amd import _lib                                              # noqa: E402
from diner_amd.synthetic import make_scene, make_mlp_state_dict, build_modules   # noqa: E402
from diner_amd import ops                                               # noqa: E402,F401

ap = argparse.ArgumentParser()
ap.add_argument("--objects", type=int, default=4)
ap.add_argument("--rays", type=int, nargs="+", default=[4096])
ap.add_argument("--samples", type=int, default=40)
ap.add_argument("--size", default="400x300")
ap.add_argument("--steps", type=int, default=5)
args = ap.parse_args()
W, H = (int(v) for v in args.size.split("x"))
dev = torch.device("cuda", 0)
SB, K = args.objects, args.samples
G = int(15 * K / 40)
scs = [make_scene(W, H, seed=s) for s in range(SB)]
nerf, R = build_modules(scs, make_mlp_state_dict(), dev)
nerf.train()
from diner_amd.synthetic import as_encoded
nerf.encoder.latent = as_encoded(nerf.encoder.latent.detach()).requires_grad_(True)      # channels-last strides, as PixelNeRF.encode emits it
lib = _lib.load()
E = torch.stack([s["target_extrinsics"] for s in scs])
Km = torch.stack([s["target_intrinsics"] for s in scs])
rays_all = ops.gen_rays(E, Km, W, H, scs[0]["znear"], scs[0]["zfar"], dev)         # (SB, H*W, 8)
for NR in args.rays:
    side = int(round(NR ** 0.5))
    if side * side == NR and side <= min(W, H):          # a side x side patch around the image centre (diner.py:233-247)
        ys, xs = torch.meshgrid(torch.arange(side) + (H - side) // 2, torch.arange(side) + (W - side) // 2, indexing="ij")
        idx = (ys * W + xs).reshape(-1)
    else:                                                # w_vgg = 0: random pixels (diner.py:231)
        idx = torch.randint(0, H * W, (NR,), generator=torch.Generator().manual_seed(0))
    r = rays_all[:, idx.to(dev)].contiguous()
    gt = torch.rand(SB, NR, 3, device=dev)
    ren = R(n_samples=K, n_depth_candidates=1000, n_gaussian=G, white_bkgd=True)

    mlp_params = [p for p in nerf.parameters()]

    def step():
        for p in nerf.parameters():
            p.grad = None
        nerf.encoder.latent.grad = None
        with torch.no_grad():        # an optimiser step's in-place write: the parameters are "new" every step, as in training -- whatever is
            torch._foreach_add_(mlp_params, 0.0)      # cached per parameter version (packed weights, projected latent maps) is redone per step
        out = ren.forward(nerf, r)
        torch.nn.functional.mse_loss(out.fine.rgb, gt).backward()

    torch.cuda.reset_peak_memory_stats()
    step()
    torch.cuda.synchronize()
    t = time.perf_counter()
    n = args.steps
    for _ in range(n):
        step()
    host = (time.perf_counter() - t) / n                   # host time to enqueue a step (the device is still working)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / n
    flop = 3 * 2 * SB * NR * K * (4 * (55 * 512 + 9 * 512 * 512) + 4 * 512 * 512 + 4 * 512)      # fwd + dgrad + wgrad
    from diner_amd import train as _train
    ws, scratch_b = _train.workspace_split(NR * K, 4)
    print(f"{SB} object(s) x {NR} rays x {K} samples: {dt * 1e3:.2f} ms per forward+backward step = {SB * NR / dt:.0f} rays/s, "
          f"{flop / dt / 1e12:.1f} TFLOP/s fp32-equivalent; host enqueue {host * 1e3:.2f} ms per step; saved activations {ws / 2 ** 30:.2f} GiB per object "
          f"({SB * ws / 2 ** 30:.1f} GiB alive between forward and backward) + {scratch_b / 2 ** 30:.2f} GiB of shared work buffers, peak device memory {torch.cuda.max_memory_allocated() / 2 ** 30:.1f} GiB",
          flush=True)
