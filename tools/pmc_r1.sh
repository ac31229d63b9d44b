#!/bin/bash
# PMC passes for the field kernels (separate passes per counter group; kernel-trace only, as gpurun requires).
# usage: tools/pmc_r1.sh <outdir-under-gpurun_out>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$1
mkdir -p $OUT
CMD="python bench.py --steps 1 --warmup 0 --cpu-rays 0 --width 200 --height 150"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_UNALIGNED_STALL TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -o pmc -- $CMD > $OUT/p$i.log 2>&1
  python - <<PY
import csv, glob, collections
f = glob.glob("$OUT/p$i/*counter_collection.csv")
if not f:
    print("pass $i: no counter file", glob.glob("$OUT/p$i/*"))
else:
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].split("(")[0]
        if "k_field" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        n[(k, r["Counter_Name"])] += 1
    for k in acc:
        for c, v in acc[k].items():
            print(f"{k:28s} {c:32s} total {v:.6e}  launches {n[(k,c)]}  per-launch {v/n[(k,c)]:.6e}")
PY
done
