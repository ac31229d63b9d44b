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


# ---- production noise: one key per frame ---------------------------------------------------------------------------------------
# The in-kernel Philox noise of ray i of a call is keyed by (seed, ray_index0 + i).  A caller that renders a frame in batches or
# shards (diner_amd.render.predict_image, bench.py) sets ONE seed for the frame and tells every renderer.forward call where its
# rays sit in the frame's ray list; the frame then does not depend on the batching.  Without a key each call draws its own seed
# from torch's global CPU generator (the reference's behaviour: noise depends on the call sequence).
def frame_key():
    return getattr(_state, "key", None)


@contextlib.contextmanager
def keyed(seed, ray_index0=0):
    prev = frame_key()
    _state.key = (int(seed), int(ray_index0))
    try:
        yield
    finally:
        _state.key = prev
