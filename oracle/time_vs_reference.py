"""CPU-baseline fidelity (SURVEY.md section 8d "CPU baseline timing"): bench.py's `cpu_baseline` times the
ORACLE (oracle/diner_oracle.py, the restatement that can travel to the GPU box); this script shows, in the build container where the
reference can be imported, that the restatement's throughput stands for the reference's:

    python oracle/time_vs_reference.py [--rays 1024] [--repeats 4] [--threads N]

The imported reference's NeRFRendererDGS.forward (nerf_renderer.py:399-424) and diner_oracle.render run on the SAME 1024 rays of the 400x300
bench scene (K = 128, 48 gaussian, 1000 candidates, injected noise), alternating, `--repeats` times each after one warm-up of each; prints
and writes (profiles/r06_cpu_oracle_vs_reference_timing.txt) every repeat, the medians, the spread and the ratio.  Build container only
(test infrastructure: imports /root/reference)."""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import diner_oracle as O                                     # noqa: E402
from oracle.ref_import import import_reference                           # noqa: E402
from oracle.make_golden import inject_noise                              # noqa: E402
from oracle.make_golden_r2 import setup, lattice                         # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=1024)
    ap.add_argument("--repeats", type=int, default=4)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r06_cpu_oracle_vs_reference_timing.txt"))
    a = ap.parse_args()
    if a.threads:
        torch.set_num_threads(a.threads)
    ns = import_reference()
    W, H, K, G, n_cand = 400, 300, 128, 48, 1000
    sc, nerf, scene, w, rays = setup(ns, W, H, 0)
    side = int(round(a.rays ** 0.5))
    rs = rays[lattice(W, H, side)].contiguous()
    NR = rs.shape[0]
    g = torch.Generator().manual_seed(5)
    nc, ng, nf = torch.rand(NR, n_cand, generator=g), torch.randn(NR, G, generator=g), torch.rand(NR, K, generator=g)
    ren = ns.nerf_renderer.NeRFRendererDGS(n_samples=K, n_depth_candidates=n_cand, n_gaussian=G, white_bkgd=False)

    def run_ref():
        t = time.perf_counter()
        with torch.no_grad(), inject_noise(nc, ng, nf):
            out = ren.forward(nerf, rs[None])
        return time.perf_counter() - t, out.fine.rgb[0]

    def run_oracle():
        t = time.perf_counter()
        with torch.no_grad():
            out = O.render(scene, w, rs, K, n_cand, G, False, nc, ng, nf)
        return time.perf_counter() - t, out["rgb"]

    _, a_rgb = run_ref()
    _, b_rgb = run_oracle()
    same = bool(torch.equal(a_rgb, b_rgb))
    tr, to = [], []
    for _ in range(a.repeats):
        tr.append(run_ref()[0])
        to.append(run_oracle()[0])
    med = lambda v: sorted(v)[len(v) // 2]
    lines = [f"{NR} rays of the {W}x{H} bench scene, K = {K}, {G} gaussian, {n_cand} candidates; torch threads {torch.get_num_threads()} of {os.cpu_count()} "
             f"hardware threads of the build container; one warm-up each, then {a.repeats} alternating repeats; outputs bit-equal: {same}",
             "imported reference (NeRFRendererDGS.forward): " + ", ".join(f"{NR / t:.1f}" for t in tr) + f" rays/s  (median {NR / med(tr):.1f}, "
             f"min {NR / max(tr):.1f}, max {NR / min(tr):.1f})",
             "oracle (diner_oracle.render):                 " + ", ".join(f"{NR / t:.1f}" for t in to) + f" rays/s  (median {NR / med(to):.1f}, "
             f"min {NR / max(to):.1f}, max {NR / min(to):.1f})",
             f"oracle / reference (medians): {med(tr) / med(to):.3f}"]
    print("\n".join(lines))
    with open(a.out, "w") as f:
        f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
