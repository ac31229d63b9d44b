#!/bin/bash
# MFMA utilisation + effective clock of the field kernels: PMC pass (kernel-trace only) + kernel durations.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-pmc_mfma}; mkdir -p $OUT
CMD="python bench.py --steps 1 --warmup 1 --cpu-rays 0 --width 256 --height 256"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $OUT/p -o pmc -- $CMD > $OUT/p.log 2>&1
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); dur=collections.defaultdict(float); n=collections.Counter()
for r in csv.DictReader(open(glob.glob("$OUT/p/*counter_collection.csv")[0])):
    k = r["Kernel_Name"].split("(")[0]
    if "k_field" in k: acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for r in csv.DictReader(open(glob.glob("$OUT/p/*kernel_trace.csv")[0])):
    k = r["Kernel_Name"].split("(")[0]
    if "k_field" in k: dur[k] += (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))*1e-9; n[k]+=1
for k in acc:
    g = acc[k]["GRBM_GUI_ACTIVE"]/8.0
    print(f"{k}: launches {n[k]} time {dur[k]*1e3:.1f} ms  clock {g/dur[k]/1e9:.3f} GHz  MfmaUtil {acc[k]['SQ_VALU_MFMA_BUSY_CYCLES']/(g*1024):.3f}")
PY
