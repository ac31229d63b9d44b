#!/bin/bash
# quick PMC comparison of the per-view kernel: tools/pmc_quick.sh <tag>  (env DINER_AMD_LIB / DINER_AMD_FIELD_KERNEL select the build)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmcq_$1; mkdir -p $OUT
CMD="python bench.py --steps 1 --warmup 1 --cpu-rays 0 --no-modes --width 256 --height 256"
for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" "TA_BUSY_avr TA_TA_BUSY_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  n=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_$n -o pmc -- $CMD > $OUT/pmc_$n.log 2>&1
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("$OUT/pmc_*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "k_field_pre" in k and ("h3n" in k or "h3w" in k): acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k,v in acc.items():
    print("$1", k)
    print("   " + "  ".join(f"{c}={x:.3e}" for c,x in sorted(v.items())))
PY
