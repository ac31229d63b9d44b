"""Explicit-noise injection for the renderer (parity tests).

The reference draws its sampling noise from torch's global generator with data-dependent shapes
(nerf_renderer.py:57, :188, :390).  In production the HIP sampler draws the same three noise fields from an in-kernel
counter-based Philox generator keyed by a seed taken from torch's global CPU generator (so torch.manual_seed
controls it).  For bit-level comparisons against the reference the three fields can be injected instead:

    with diner_amd.noise.inject(coarse=(SB,NR,n_cand), gauss=(SB,NR,G), fill=(SB,NR,K)):
        renderer.forward(model, rays)
"""
import contextlib
import threading

_state = threading.local()


def current():
    return getattr(_state, "noise", None)


@contextlib.contextmanager
def inject(coarse, gauss, fill):
    prev = current()
    _state.noise = (coarse, gauss, fill)
    try:
        yield
    finally:
        _state.noise = prev
