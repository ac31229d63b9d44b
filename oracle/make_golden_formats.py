"""Fixtures for diner_amd/formats.py (row f4): small PFM files in the variants the format allows and what the
REFERENCE's reader (src/util/io.py:4-39, importable: numpy only) returns for them.  Run here (needs /root/reference):
    python oracle/make_golden_formats.py
Test infrastructure only."""
import importlib.util
import os
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "formats")
spec = importlib.util.spec_from_file_location("ref_io", "/root/reference/src/util/io.py")
ref_io = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref_io)

rng = np.random.default_rng(7)
os.makedirs(OUT, exist_ok=True)
cases = {"grey_le": ("Pf", "<", (5, 7)), "grey_be": ("Pf", ">", (4, 6)), "color_le": ("PF", "<", (3, 5, 3))}
expected = {}
for name, (hdr, endian, shape) in cases.items():
    a = rng.normal(size=shape).astype(np.float32)
    path = os.path.join(OUT, name + ".pfm")
    with open(path, "wb") as f:
        f.write(f"{hdr}\n{shape[1]} {shape[0]}\n{'-2.5' if endian == '<' else '0.5'}\n".encode())
        f.write(a.astype(endian + "f4").tobytes())
    data, scale = ref_io.read_pfm(path)
    expected[name] = np.ascontiguousarray(data, dtype=np.float32)
    expected[name + "_scale"] = np.float64(scale)
np.savez(os.path.join(OUT, "pfm_expected.npz"), **expected)
print("wrote", sorted(os.listdir(OUT)))
