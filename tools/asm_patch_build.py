"""Debug aid: rebuild libdiner_hip_<name>.so from an edited device assembly of mlp_h3n.hip.
usage: asm_patch_build.py NAME MODE [LO HI]   MODE: none | vm0 | lgkm0 | all0, applied to s_waitcnt lines LO..HI (1-based)"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
FLAGS = "--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-value -Wno-comment".split()
SRC = os.path.join(ROOT, "diner_amd/csrc/mlp_h3n.hip")
name, mode = sys.argv[1], sys.argv[2]
lo, hi = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1, 10**9)
extra = sys.argv[5:]   # extra -D flags
work = os.path.join(ROOT, "build", "asm_" + name); os.makedirs(work, exist_ok=True)
base_s = os.path.join(ROOT, "build", "asm_base.s")
if not os.path.exists(base_s) or os.environ.get("REGEN"):
    subprocess.check_call(["hipcc"] + FLAGS + extra + ["-x", "hip", "-S", "--cuda-device-only", SRC, "-o", base_s],
                          stderr=subprocess.DEVNULL)
out = []
n = 0
for i, line in enumerate(open(base_s), 1):
    m = re.match(r"\s*s_waitcnt\s+(.*)", line)
    if m and lo <= i <= hi and mode != "none" and "_depctr" not in line:
        ops = m.group(1)
        vm = re.search(r"vmcnt\((\d+)\)", ops); lg = re.search(r"lgkmcnt\((\d+)\)", ops)
        parts = []
        if vm: parts.append("vmcnt(0)" if mode in ("vm0", "all0") else vm.group(0))
        if lg: parts.append("lgkmcnt(0)" if mode in ("lgkm0", "all0") else lg.group(0))
        ex = re.search(r"expcnt\((\d+)\)", ops)
        if ex: parts.append(ex.group(0))
        if mode == "dec" and vm:
            parts = [f"vmcnt({max(int(vm.group(1)) - 1, 0)})"] + ([lg.group(0)] if lg else [])
        if mode == "full":
            parts = ["vmcnt(0)", "lgkmcnt(0)"]
        if parts:
            line = "\ts_waitcnt " + " ".join(parts) + "\n"; n += 1
    if mode == "ins" and lo <= i <= hi and re.match(r"\s+[vsdg][a-z_]*_[a-z0-9_]+\s", line) and not line.strip().startswith(("s_waitcnt", "s_endpgm", "s_branch", "s_cbranch", "s_setpc", "s_getpc")):
        out.append("\ts_waitcnt vmcnt(0) lgkmcnt(0)\n"); n += 1
    out.append(line)
    if mode == "spillfix" and re.match(r"\s+scratch_store_dwordx?\d* off, a\[", line):
        out.append("\ts_waitcnt vmcnt(0)\n"); n += 1
mod_s = os.path.join(work, "dev.s"); open(mod_s, "w").writelines(out)
dev_o, dev_out, fb, host_o = [os.path.join(work, f) for f in ("dev.o", "dev.out", "dev.hipfb", "host.o")]
subprocess.check_call([f"{LLVM}/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", mod_s, "-o", dev_o])
subprocess.check_call([f"{LLVM}/lld", "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-o", dev_out, dev_o])
subprocess.check_call([f"{LLVM}/clang-offload-bundler", "-type=o", "-bundle-align=4096",
                       "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950", "-input=/dev/null",
                       f"-input={dev_out}", f"-output={fb}"])
subprocess.check_call(["hipcc"] + FLAGS + extra + ["--cuda-host-only", "-Xclang", "-fcuda-include-gpubinary", "-Xclang", fb,
                                           "-x", "hip", "-c", SRC, "-o", host_o], stderr=subprocess.DEVNULL)
objs = [os.path.join(ROOT, "build/obj", f) for f in os.listdir(os.path.join(ROOT, "build/obj")) if f.endswith(".o") and "mlp_h3n" not in f]
lib = os.path.join(ROOT, "diner_amd", f"libdiner_hip_{name}.so")
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + [host_o])
print(f"{name}: {n} waits edited ({mode}, lines {lo}..{hi}) -> {lib}")
