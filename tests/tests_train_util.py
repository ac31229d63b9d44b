"""Helpers of the training-path tests: the oracle's MLPWeights field <-> ResnetFC parameter name mapping."""
import torch


def oracle_key(k):
    """ResnetFC parameter name -> (MLPWeights field, index)."""
    parts = k.split(".")
    kind = "w" if parts[-1] == "weight" else "b"
    if parts[0] in ("lin_in", "lin_out"):
        return (f"{parts[0]}_{kind}", None)
    if parts[0] == "lin_z":
        return (f"lin_z_{kind}", int(parts[1]))
    return (f"fc{parts[2][-1]}_{kind}", int(parts[1]))


def module_param_list(msd):
    """state dict of src.models.resnetfc.ResnetFC -> (30 leaf tensors on the GPU in diner_amd.train.PARAM_ORDER,
    the matching (MLPWeights field, index) keys of the oracle)."""
    from diner_amd.train import PARAM_ORDER
    params, names = [], []
    for k in PARAM_ORDER:
        params.append(msd[k].detach().clone().cuda().requires_grad_(True))
        parts = k.split(".")
        kind = "w" if parts[-1] == "weight" else "b"
        if parts[0] == "lin_in":
            names.append((f"lin_in_{kind}", None))
        elif parts[0] == "lin_out":
            names.append((f"lin_out_{kind}", None))
        elif parts[0] == "lin_z":
            names.append((f"lin_z_{kind}", int(parts[1])))
        else:
            names.append((f"fc{parts[2][-1]}_{kind}", int(parts[1])))
    return params, names


def saved_relu_masks(out, P, nv=4, obj=0, n_obj=1):
    """The relu decisions of the HIP training forward that produced `out` (diner_amd.train.field_train / field_train_batch): signs of the
    pre-activations it saved in its workspace (diner_field_train_ws_layout) -> the relu_masks dict of oracle.diner_oracle.mlp_forward (CPU
    bool tensors).  A batched node (ABI v6) keeps the n_obj objects' rows object-major in one workspace laid out for n_obj * P points:
    `obj` selects the object."""
    import ctypes as C
    from diner_amd import _lib
    lib = _lib.load()
    ws = out.grad_fn.saved_tensors[0]
    off = (C.c_longlong * 12)()
    _lib.check(lib.diner_field_train_ws_layout(P * n_obj, nv, off, 12))
    wf = ws.view(torch.float32)

    def grab(o, rows, cols=512):
        o = o + obj * rows * cols
        return (wf[o:o + rows * cols].view(rows, cols) > 0).cpu()
    X = [grab(off[b], P * nv).view(nv, P, 512) if b < 3 else grab(off[b], P) for b in range(5)]
    H = [grab(off[5 + b], P * nv).view(nv, P, 512) if b < 3 else grab(off[5 + b], P) for b in range(5)]
    raw = grab(off[11], P, 4)
    return dict(X=X, H=H, last=grab(off[10], P), sigma=raw[:, 3:4])
