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
