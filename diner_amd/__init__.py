"""diner_amd: MI355X-native (gfx950) implementation of DINER's volumetric-rendering hot path.

The compute lives in libdiner_hip.so (hand-written HIP kernels behind the C ABI of include/diner_hip.h);
this package is the thin Python host: ctypes bindings (`_lib`), torch-facing wrappers (`ops`), the image
harness with ray sharding across GPUs (`render`) and the synthetic scenes used by tests and bench.py.
There is no CPU or eager-PyTorch fallback for the hot path: importing `diner_amd.ops` fails loudly when
the shared library has not been built (python -m diner_amd.build).
"""
__version__ = "0.1.0"
