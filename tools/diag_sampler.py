"""Diagnostic (GPU box): where does the HIP sampler deviate from the oracle on the golden sampler fixture?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from oracle import diner_oracle as O
from tests.helpers import load, oracle_setup
from tests.test_hip_parity import _sampler_case, hip_scene, hip_mlp, T
from diner_amd import ops

for K in (64, 128):
    g, sc, scene, rs, (nc, ng, nf) = _sampler_case(K)
    G = int(g["G"])
    hs = hip_scene(ops, sc)
    z, zu = ops.sample_depthguided(hs, rs.cuda(), K, 1000, G, 0.05, noise=(nc.cuda(), ng.cuda(), nf.cuda()), want_unfilled=True)
    zu = zu.cpu()
    z0, aux = O.sample_depthguided(scene, rs, K, 1000, G, nc, ng, return_aux=True)
    L, zc = aux["L"], aux["z_cand"]
    ref_u, got_u = z0.sort(-1).values, zu.sort(-1).values
    bad = (~torch.isclose(got_u, ref_u, rtol=3e-6, atol=1e-7).all(-1)).nonzero().flatten()
    print(f"K={K}: {len(bad)} rays differ; positive-L counts of those rays:", (L[bad] > 0).sum(-1).tolist(), "want", K - G)
    for r in bad[:8].tolist():
        sref, sgot = set(z0[r, :K - G].tolist()), set(zu[r, :K - G].tolist())
        only_ref = sorted(sref - sgot); only_got = sorted(sgot - sref)
        def Lof(zs):
            out = []
            for zz in zs:
                i = (zc[r] == zz).nonzero().flatten()
                out.append(float(L[r, i[0]]) if len(i) else None)
            return out
        print(f"  ray {r}: only in ref {only_ref} L={Lof(only_ref)}; only in hip {only_got} L={Lof(only_got)}")
        Ls = L[r].sort(descending=True).values
        print(f"     L around cut-off: {Ls[K-G-3:K-G+3].tolist()}")
