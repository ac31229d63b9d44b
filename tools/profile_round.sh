#!/bin/bash
# Round profile: rocprofv3 kernel stats of the bench command + PMC passes (MFMA utilisation, HBM traffic).
# usage (GPU box): tools/profile_round.sh <tag>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${1:-r01}; OUT=gpurun_out/prof_$TAG; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python bench.py --steps 2 --warmup 1 --cpu-rays 0 > $OUT/bench_stats.log 2>&1
grep '"metric"' $OUT/bench_stats.log > $OUT/bench_line.json
CMD="python bench.py --steps 1 --warmup 1 --cpu-rays 0"
for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_$n -o pmc -- $CMD > $OUT/pmc_$n.log 2>&1
done
python - <<PY
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(float)); dur=collections.defaultdict(float); n=collections.Counter()
for f in glob.glob("$OUT/pmc_*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "diner::" in k: acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
f = glob.glob("$OUT/pmc_SQ_VALU_MFMA_BUSY_CYCLES/*kernel_trace.csv")[0]
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0]
    if "diner::" in k: dur[k] += (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))*1e-9; n[k]+=1
out = {}
for k in acc:
    g = acc[k].get("GRBM_GUI_ACTIVE", 0)/8.0
    d = dict(launches=n[k], time_ms=round(dur[k]*1e3,3), counters={c: v for c, v in acc[k].items()})
    if g and dur[k]:
        d["clock_GHz"] = round(g/dur[k]/1e9,3); d["MfmaUtil"] = round(acc[k].get("SQ_VALU_MFMA_BUSY_CYCLES",0)/(g*1024),4)
    if "FETCH_SIZE" in acc[k]:
        d["hbm_read_GB_x2_corrected"] = round(acc[k]["FETCH_SIZE"]*2*1024/1e9,3); d["hbm_write_GB"] = round(acc[k].get("WRITE_SIZE",0)*1024/1e9,3)
    out[k] = d
json.dump(out, open("$OUT/pmc_summary.json","w"), indent=1)
print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "counters"} for k, v in out.items()}, indent=1))
PY
cat $OUT/bench_line.json | cut -c1-200
