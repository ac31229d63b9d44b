#!/bin/bash
# PMC passes for the kernels whose name contains <substring>: tools/pmc_kernel.sh <tag> <substring> <command ...>   (GPU box)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=$1; SUB=$2; shift 2
OUT=gpurun_out/pmck_$TAG; mkdir -p $OUT
for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  n=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_$n -o pmc -- "$@" > $OUT/pmc_$n.log 2>&1
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("$OUT/pmc_*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "$SUB" in k: acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in acc.items():
    print("$TAG", k)
    for c, x in sorted(v.items()): print(f"   {c:34s} {x:.4e}")
    g = v.get("GRBM_GUI_ACTIVE", 0) / 8
    if g: print(f"   MfmaUtil {v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (g * 1024):.3f}")
PY
