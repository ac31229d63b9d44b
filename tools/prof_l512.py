"""Phase timer of lin512_body (the 512 x 512 layer products of the training path, csrc/train_lin512.hip):

    python -c "from diner_amd.build import build_variant; build_variant('l512prof', ['DINER_L512_PROF'])"
    DINER_AMD_LIB=diner_amd/libdiner_hip_l512prof.so python tools/prof_l512.py [rows]

Shader clocks per wave, summed by the kernel: the MFMA slab loops, the slab barriers, the epilogues.  (1) the forward product in f16x3 at
`rows` rows (128-row tiles), plain / with a residual / accumulating; (2) one object-step of the training path (every lin512 launch of it)."""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from diner_amd import _lib, train                                      # noqa: E402

lib = _lib.load()
raw = C.CDLL(_lib.LIB_PATH)
raw.diner_debug_l512_prof.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]


def read(tag, ms=None):
    a = (C.c_ulonglong * 8)()
    raw.diner_debug_l512_prof(a, 1)
    tot, slab, bar, epi, tiles, waves = (float(a[i]) for i in range(6))
    if not waves:
        print(tag, "no lin512 launches")
        return
    mf = 49152.0 * tiles / 1.0      # MFMA clocks of a 128-row f16x3 tile per wave: 4 slabs x 8 steps x 12 groups x 4 MFMAs x 32 clocks
    print(f"{tag}: waves {waves:.0f} tiles/wave {tiles / waves:.1f}  clocks/tile {tot / tiles:.0f} = slabs {slab / tiles:.0f} + barriers {bar / tiles:.0f} + "
          f"epilogue {epi / tiles:.0f} + rest {(tot - slab - bar - epi) / tiles:.0f}   (MFMA issue alone 49152 at 128-row tiles: {mf / tot:.3f} of the clocks)"
          + (f"  {ms:.3f} ms" if ms else ""), flush=True)


dev = torch.device("cuda", 0)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 655360
x = torch.randn(M, 512, device=dev)
W = torch.randn(512, 512, device=dev) * 0.04
y = torch.empty(M, 512, device=dev)
res = torch.randn(M, 512, device=dev)
b = torch.randn(512, device=dev)


def run(tag, **kw):
    train.linear512(x, W, y, f16x3=True, bias=b, **kw)
    torch.cuda.synchronize()
    read("warm-up " + tag)
    t = time.perf_counter()
    for _ in range(5):
        train.linear512(x, W, y, f16x3=True, bias=b, **kw)
    torch.cuda.synchronize()
    read(tag, (time.perf_counter() - t) / 5 * 1e3)


run("forward f16x3 plain")
run("forward f16x3 + residual", resid=res)
run("forward f16x3 relu_in", relu_in=True)
run("forward f16x3 accumulating", accumulate=True)
run("forward f16x3 fp32 mask + accumulating", accumulate=True, mask=res)
for Msmall in (163840, 20480):
    x, y, res = x[:Msmall], y[:Msmall], res[:Msmall]
    run(f"forward f16x3 plain, {Msmall} rows")

# ---- (round 6) one training step (1 object x 4096 rays x 40 samples): every lin512 launch of it, and the eight-wave weight gradient
# (k_wgrad512_w8: slab sections against slab barriers, per wave)
try:
    raw.diner_debug_wgrad_prof.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    from diner_amd import ops
    from diner_amd.synthetic import make_scene, make_mlp_state_dict, build_modules
    del x, y, res
    torch.cuda.empty_cache()
    Wt, Ht, NR, K = 400, 300, 4096, 40
    scs = [make_scene(Wt, Ht, seed=0)]
    nerf, R = build_modules(scs, make_mlp_state_dict(), dev)
    nerf.train()
    from diner_amd.synthetic import as_encoded
    nerf.encoder.latent = as_encoded(nerf.encoder.latent.detach()).requires_grad_(True)      # channels-last strides, as PixelNeRF.encode emits it
    rays_all = ops.gen_rays(torch.stack([scs[0]["target_extrinsics"]]), torch.stack([scs[0]["target_intrinsics"]]), Wt, Ht, scs[0]["znear"], scs[0]["zfar"], dev)
    ys, xs = torch.meshgrid(torch.arange(64) + (Ht - 64) // 2, torch.arange(64) + (Wt - 64) // 2, indexing="ij")
    r = rays_all[:, (ys * Wt + xs).reshape(-1).to(dev)].contiguous()
    gt = torch.rand(1, NR, 3, device=dev)
    ren = R(n_samples=K, n_depth_candidates=1000, n_gaussian=15, white_bkgd=True)
    a8 = (C.c_ulonglong * 8)()
    for rep in range(4):
        for p in nerf.parameters():
            p.grad = None
        nerf.encoder.latent.grad = None
        torch.nn.functional.mse_loss(ren.forward(nerf, r).fine.rgb, gt).backward()
        torch.cuda.synchronize()
        read(f"training step ({'warm-up' if rep == 0 else 'second'}): all lin512 launches")
        raw.diner_debug_wgrad_prof(a8, 1)
        tot, mf, bar, slabs, waves = (float(a8[i]) for i in range(5))
        if waves:
            print(f"k_wgrad512_w8 over the step: waves {waves:.0f}, slabs/wave {slabs / waves:.0f}; clocks per 32-row slab {tot / slabs:.0f} = slab section "
                  f"{mf / slabs:.0f} + barrier {bar / slabs:.0f} + rest {(tot - mf - bar) / slabs:.0f}  (MFMA issue alone: 48 x 32 = 1536 clocks per wave and slab, two waves per SIMD: 3072)",
                  flush=True)
        try:      # the merged latent-gradient scatter (k_scatter_latent_merged): thread 0's clocks per phase
            raw.diner_debug_scatter_prof.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
            raw.diner_debug_scatter_prof(a8, 2 if rep == 1 else 1)      # third step: the scatter without its atomics (wrong gradients, a measurement)
            n = float(a8[4])
            if n:
                print(("(no atomics) " if rep == 2 else "") + "k_scatter_latent_merged over the step: %.0f workgroups; clocks per workgroup: issue loads %.0f, wait for the block %.0f, "
                      "leaders / sort %.0f, sums + atomics %.0f" % (n, a8[0] / n, a8[1] / n, a8[2] / n, a8[3] / n), flush=True)
        except AttributeError:
            pass
except AttributeError:
    pass
