"""Which HIP runtime(s) end up in the process when libdiner_hip.so is loaded before / after torch (smoke() vs pytest order)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
order = sys.argv[1] if len(sys.argv) > 1 else "lib_first"
if order == "lib_first":
    from diner_amd import _lib
    _lib.load()
    import torch
else:
    import torch
    from diner_amd import _lib
    _lib.load()
print("order", order, "cuda available", torch.cuda.is_available())
x = torch.zeros(4, device="cuda")
maps = sorted({l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l or "libhsa-runtime" in l})
print("\n".join(maps))
from diner_amd import ops
from diner_amd.synthetic import make_mlp_state_dict
try:
    m = ops.HipMlp({k: v.cuda() for k, v in make_mlp_state_dict().items()})
    print("HipMlp ok", m.wmax)
except Exception as e:
    print("HipMlp FAILED:", e)
