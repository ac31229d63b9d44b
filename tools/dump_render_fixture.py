"""GPU: run the HIP sampler + renderer on the G9 / G10 fixtures and dump z / rgb / depth for analysis against the oracle."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diner_amd import ops
from tests.test_hip_parity import _render_fixture, hip_scene, hip_mlp, T
for name in ("g9_render_K128", "g10_render_cfg5"):
    g, sc, scene, w, msd, (K, G, n_cand, white), (nc, ng, nf) = _render_fixture(name)
    hs, hm = hip_scene(ops, sc), hip_mlp(ops, msd)
    rc = T(g["rays"]).cuda()
    z, zu = ops.sample_depthguided(hs, rc, K, n_cand, G, 0.05, noise=(nc.cuda(), ng.cuda(), nf.cuda()), want_unfilled=True)
    out = {}
    for mode in ("f16x3", "fp32"):
        _, rgb, depth = ops.render(hs, hm, rc, z, white, precision=mode)
        out["rgb_" + mode], out["depth_" + mode] = rgb.cpu().numpy(), depth.cpu().numpy()
    np.savez_compressed(os.path.join(ROOT, "gpurun_out", f"dump_{name}.npz"), z=z.cpu().numpy(), zu=zu.cpu().numpy(), **out)
    print(name, "dumped")
