#!/bin/bash
# MfmaUtil and held clock of the field kernels on the headline workload: tools/pmc_util.sh <tag>   (env DINER_AMD_LIB selects the build)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmcu_$1; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc -o pmc -- python bench.py --steps 1 --warmup 0 --cpu-rays 0 --no-modes --no-configs > $OUT/pmc.log 2>&1
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); dur = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob("$OUT/pmc/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "h3n::k_field" in k: acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for f in glob.glob("$OUT/pmc/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "h3n::k_field" in k: dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9; n[k] += 1
for k, v in acc.items():
    g = v["GRBM_GUI_ACTIVE"] / 8.0
    print("$1", k, "launches %d  avg %.3f ms  clock %.3f GHz  MfmaUtil %.4f" % (n[k], dur[k] / n[k] * 1e3, g / dur[k] / 1e9, v["SQ_VALU_MFMA_BUSY_CYCLES"] / (g * 1024)))
PY
