"""Image output on the device + a dependency-free PNG writer.

DINER.create_prediction_folder (diner.py:119-133) colour-maps the depth with `torch_cmap` (torch_helpers.py:42-75:
matplotlib viridis, per-image min / max, float64) and writes everything with torchvision's `save_image`
(uint8(clamp(v * 255 + 0.5, 0, 255))).  Here the per-pixel work (quantisation, min / max, colour lookup) runs in HIP
kernels on the rendered tensors, and the PNG container is written with zlib only (PIL / torchvision are not needed).
"""
import struct
import zlib

import numpy as np
import torch

from . import _lib
from .ops import _ptr, _stream, _require_hip, _f32c

lib = _lib.load()
_luts = {}


def _lut_u8(cmap, device):
    """256 x 3 uint8 table of a matplotlib colormap, quantised exactly as save_image quantises torch_cmap's float64 output."""
    key = (cmap, str(device))
    if key not in _luts:
        import matplotlib.pyplot as plt
        c = plt.get_cmap(cmap)
        rgb = c(np.arange(c.N))[:, :3]                                             # float64, as cmap(x) returns
        _luts[key] = torch.from_numpy(np.clip(rgb * 255 + 0.5, 0, 255).astype(np.uint8)).contiguous().to(device)
    return _luts[key]


def to_uint8(img):
    """save_image's quantisation on the device: img (3,H,W) float32 in [0,1] -> (H,W,3) uint8."""
    _require_hip(img)
    img = _f32c(img)
    assert img.dim() == 3 and img.shape[0] == 3
    _, H, W = img.shape
    out = torch.empty(H, W, 3, dtype=torch.uint8, device=img.device)
    with torch.cuda.device(img.device):
        _lib.check(lib.diner_quantize_rgb_u8(_ptr(img), H, W, _ptr(out), _stream()))
    return out


def _key_to_float(k):
    b = k if k >= 0 else k ^ 0x7fffffff
    return struct.unpack("<f", struct.pack("<i", b))[0]


def depth_to_uint8(depth, cmap="viridis", vmin=None, vmax=None):
    """torch_cmap + save_image for one depth map: depth (H,W) or (1,H,W) float32 -> (H,W,3) uint8.  Like the reference,
    a vmin / vmax of None *or 0* means the image's own minimum / maximum (torch_helpers.py:63-64)."""
    _require_hip(depth)
    d = _f32c(depth).reshape(depth.shape[-2], depth.shape[-1])
    H, W = d.shape
    with torch.cuda.device(d.device):
        if not vmin or not vmax:
            mm = torch.empty(2, dtype=torch.int32, device=d.device)
            _lib.check(lib.diner_minmax_f32(_ptr(d), H * W, _ptr(mm), _stream()))
            lo, hi = (_key_to_float(int(v)) for v in mm.cpu().tolist())
            vmin = vmin if vmin else lo
            vmax = vmax if vmax else hi
        out = torch.empty(H, W, 3, dtype=torch.uint8, device=d.device)
        _lib.check(lib.diner_colormap_u8(_ptr(d), H * W, _ptr(_lut_u8(cmap, d.device)), float(vmin), float(vmax),
                                         _ptr(out), _stream()))
    return out


from .png import write_png, read_png      # noqa: E402,F401  (the container itself needs neither torch nor the extension)


def save_prediction(outdir, stem, rgb, depth, pred_suffix="_pred.png", depth_suffix="_depth.png"):
    """One target view as create_prediction_folder writes it: rgb (3,H,W), depth (1,H,W) on the device -> two PNG files."""
    import os
    os.makedirs(outdir, exist_ok=True)
    write_png(os.path.join(outdir, stem + pred_suffix), to_uint8(rgb))
    write_png(os.path.join(outdir, stem + depth_suffix), depth_to_uint8(depth))
