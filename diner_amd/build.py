"""Build libdiner_hip.so (gfx950 only) in-tree with hipcc.  No torch involved: the library is a plain
C-ABI shared object (include/diner_hip.h) that the Python host loads with ctypes.

    python -m diner_amd.build          # or: from diner_amd.build import build; build()
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(ROOT, "build", "obj")
LIB = os.path.join(HERE, "libdiner_hip.so")
SOURCES = ["api.cpp", "sampler.hip", "composite.hip", "stage_ops.hip", "prep.hip", "train.hip", "train_512.hip", "mlp.hip", "mlp_h3n.hip", "generic.hip"]
# -ffp-contract=off: every fp32 op of the geometry path rounds where the reference's torch ops round;
# fused multiply-adds are written explicitly (fmaf / MFMA) where they are wanted.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wno-unused-value",
         "-Wno-comment"]


# per-source additions: beside MFMAs packed fp32 VALU instructions (v_pk_add_f32 / v_pk_mul_f32, which -O3's SLP vectoriser forms from
# adjacent scalar operations) are an anti-lever (MI355X_MICROARCH.md, "price of one filler beside MFMAs")
EXTRA_FLAGS = {}


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _digest(paths):
    h = hashlib.sha256()
    for p in sorted(paths):
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    h.update(repr(sorted(EXTRA_FLAGS.items())).encode())
    return h.hexdigest()


def build_variant(name, defines, verbose=False):
    """Timing-experiment builds (tools/ablate.sh): libdiner_hip_<name>.so with extra -D flags."""
    obj_dir = os.path.join(ROOT, "build", "obj_" + name)
    os.makedirs(obj_dir, exist_ok=True)
    hipcc = _hipcc()
    objs = []
    for src in SOURCES:
        op = os.path.join(obj_dir, src + ".o")
        cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(src, []) + ["-D" + d for d in defines] + ["-x", "hip", "-c", os.path.join(CSRC, src), "-o", op]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        objs.append(op)
    lib = os.path.join(HERE, f"libdiner_hip_{name}.so")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    return lib


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    deps = [os.path.join(CSRC, "common.hpp"), os.path.join(CSRC, "field_common.hpp"), os.path.join(CSRC, "train_lin512.hpp"),
            os.path.join(CSRC, "train_lin512.hip"), os.path.join(CSRC, "train_wgrad512.hip"),      # included by train_512.hip
            os.path.join(ROOT, "include", "diner_hip.h")]
    hipcc = _hipcc()
    objs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        op = os.path.join(OBJ, src + ".o")
        stamp = op + ".sha"
        dig = _digest([sp] + deps)
        if force or not os.path.exists(op) or not os.path.exists(stamp) or open(stamp).read() != dig:
            cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(src, []) + ["-x", "hip", "-c", sp, "-o", op]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            with open(stamp, "w") as f:
                f.write(dig)
        objs.append(op)
    if force or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
