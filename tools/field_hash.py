"""Bit-level A/B of library builds (GPU box): sha256 of the field / render outputs on a fixed seeded input.
usage: DINER_AMD_LIB=<lib.so> python tools/field_hash.py   -> one line per precision; identical lines = bit-identical kernels."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.helpers import oracle_setup
from tests.test_hip_parity import hip_scene, hip_mlp
from diner_amd import ops

sc, scene, w, msd, rays = oracle_setup(64, 48, 7)
hs, hm = hip_scene(ops, sc), hip_mlp(ops, msd)
g = torch.Generator().manual_seed(3)
z = (0.6 + 0.8 * torch.rand(rays.shape[0], 40, generator=g)).sort(-1).values
for name, prec in (("f16x3", ops.PRECISION_F16X3), ("f16", ops.PRECISION_F16), ("fp32", ops.PRECISION_FP32)):
    f = ops.field_from_rays(hs, hm, rays.cuda(), z.cuda(), precision=prec).cpu().contiguous()
    print(name, hashlib.sha256(f.numpy().tobytes()).hexdigest()[:16], float(f.abs().sum()))
