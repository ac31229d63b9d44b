"""Timing of the per-image preparation kernels (rows f2/f3): HIP events, achieved HBM bytes/s on algorithmic bytes."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from diner_amd import ops
from diner_amd.synthetic import make_scene
def timeit(f, n=20):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for W, H in ((400, 300), (800, 600)):
    sc = make_scene(W, H, seed=0)
    d, K = sc["depths"].cuda(), sc["src_intrinsics"].cuda()
    t = timeit(lambda: ops.depth2normal(d, K))
    byts = d.numel() * 4 * (1 + 3)      # read depth once (neighbours from cache), write 3 planes
    print(f"depth2normal 4x{W}x{H}: {t*1e6:.1f} us, {byts/t/1e9:.0f} GB/s algorithmic")
    E, Km = sc["target_extrinsics"].view(1, 4, 4), sc["target_intrinsics"].view(1, 3, 3)
    t = timeit(lambda: ops.gen_rays(E, Km, W, H, 0.5, 1.5, "cuda"))
    print(f"gen_rays {W}x{H}: {t*1e6:.1f} us (incl. host packing), {W*H*32/t/1e9:.0f} GB/s written")
