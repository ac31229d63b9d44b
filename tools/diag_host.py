import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from oracle import diner_oracle as O
from tests.helpers import load, oracle_setup, selection_diff
from tests.test_oracle_golden import _sampler_inputs, T
for K in (64,128):
    g = load(f"g3_sampler_K{K}.npz")
    scene, rs, noises = _sampler_inputs(g)
    nc, ng, nf = noises[K]
    z0, aux = O.sample_depthguided(scene, rs, K, 1000, int(g["G"]), nc, ng, return_aux=True)
    bad, worst = selection_diff(T(g["z_unfilled"]).sort(-1).values, z0.sort(-1).values, aux["L"], aux["z_cand"])
    print(K, "bad rays", len(bad), "worst L", worst)
    np.testing.assert_allclose(aux["L"].sum(-1).numpy(), g["L_sum"], rtol=1e-6)
    for r in bad[:3]:
        a = T(g["z_unfilled"])[r].sort().values; b = z0[r].sort().values
        d = (a-b).abs(); print("  ray", r, "n diff", (d>1e-7).sum().item(), "max", d.max().item(), "L>0 count", (aux["L"][r]>0).sum().item())
