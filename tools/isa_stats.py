"""Per-kernel statistics of a hipcc -S listing: register counts, spills, instruction mix, spills inside the MFMA span.
usage: python tools/isa_stats.py file.s [kernel-name-substring]"""
import re, sys
txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
meta = {}
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S):
    d = dict(re.findall(r"\.amdhsa_(next_free_vgpr|next_free_sgpr|accum_offset|private_segment_fixed_size) (\d+)", m.group(2)))
    meta[m.group(1)] = d
for m in re.finditer(r"^(\S+):\s*; @\1\n(.*?)\.Lfunc_end", txt, re.S | re.M):
    name, body = m.group(1), m.group(2).split("\n")
    if pat not in name or "pack" in name:
        continue
    ins = [l.split(";")[0].strip() for l in body if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    ins = [l for l in ins if l]
    idx = [i for i, l in enumerate(ins) if l.startswith("v_mfma")]
    def cnt(p, seq=ins): return sum(1 for l in seq if l.startswith(p))
    span = ins[idx[0]:idx[-1] + 1] if idx else []
    print(name[:70])
    print("  meta", meta.get(name))
    print(f"  instr {len(ins)}  mfma {len(idx)}  global_load {cnt('global_load')}  ds_read {cnt('ds_read')}  ds_write {cnt('ds_write')} "
          f"global_store {cnt('global_store')}  s_waitcnt {cnt('s_waitcnt')}  s_nop {cnt('s_nop')}  s_barrier {cnt('s_barrier')}")
    print(f"  valu(v_ non-mfma) {sum(1 for l in ins if l.startswith('v_') and not l.startswith('v_mfma'))}  accvgpr_read {cnt('v_accvgpr_read')}  accvgpr_write {cnt('v_accvgpr_write')}")
    print(f"  scratch total {cnt('scratch_')}  inside MFMA span {cnt('scratch_', span)} (loads {cnt('scratch_load', span)}, stores {cnt('scratch_store', span)})")
