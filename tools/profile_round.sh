#!/bin/bash
# Round profile: rocprofv3 kernel stats of the bench command + PMC passes (MFMA utilisation, HBM traffic, wave states).
# usage (GPU box): tools/profile_round.sh <tag> [bench.py arguments, e.g. --width 800 --height 600]
# Writes gpurun_out/prof_<tag>/{bench_line.json, kernel_stats.md, pmc_summary.json, pmc_latest_entry.json}.
# Counters are collected in their own passes (--pmc with --kernel-trace only), one group per pass (TCC slots: FETCH_SIZE
# and WRITE_SIZE do not fit together), as /opt/skills/guides/MI355X_MICROARCH.md prescribes.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${1:-r02}; shift
ARGS="$@"
OUT=gpurun_out/prof_$TAG; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python bench.py --steps 2 --warmup 1 --cpu-rays 0 --no-modes --no-configs --no-train --no-encode --no-power $ARGS > $OUT/bench_stats.log 2>&1
grep '"metric"' $OUT/bench_stats.log > $OUT/bench_line.json
CMD="python bench.py --steps 1 --warmup 1 --cpu-rays 0 --no-modes --no-configs --no-train --no-encode --no-power $ARGS"
for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_$n -o pmc -- $CMD > $OUT/pmc_$n.log 2>&1
done
python - <<PY
import csv, glob, collections, json, sys, os
sys.path.insert(0, os.getcwd())
OUT = "$OUT"
# ---- kernel stats (timed region = 2 steps + 1 warm-up step of the bench command)
rows = []
for f in glob.glob(OUT + "/stats/*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
with open(OUT + "/kernel_stats.md", "w") as md:
    md.write("| kernel | calls | total ms | avg ms | % |\n|---|---|---|---|---|\n")
    for r in rows[:14]:
        md.write(f"| {r['Name'][:90]} | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.2f} | {float(r['AverageNs'])/1e6:.3f} | {float(r['Percentage']):.2f} |\n")
# ---- PMC
acc = collections.defaultdict(lambda: collections.defaultdict(float)); dur = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob(OUT + "/pmc_*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "diner::" in k: acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
f = glob.glob(OUT + "/pmc_SQ_VALU_MFMA_BUSY_CYCLES/*kernel_trace.csv")[0]
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0]
    if "diner::" in k: dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9; n[k] += 1
out = {}
for k in acc:
    g = acc[k].get("GRBM_GUI_ACTIVE", 0) / 8.0
    d = dict(launches=n[k], time_ms=round(dur[k] * 1e3, 3), counters={c: v for c, v in acc[k].items()})
    if g and dur[k]:
        d["clock_GHz"] = round(g / dur[k] / 1e9, 3); d["MfmaUtil"] = round(acc[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (g * 1024), 4)
    if "FETCH_SIZE" in acc[k]:
        d["hbm_read_GB_x2_corrected"] = round(acc[k]["FETCH_SIZE"] * 2 * 1024 / 1e9, 3); d["hbm_write_GB"] = round(acc[k].get("WRITE_SIZE", 0) * 1024 / 1e9, 3)
    c = acc[k]
    if c.get("SQ_WAVE_CYCLES"):
        pass
    if c.get("TCC_HIT_sum") is not None and (c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0)) > 0:
        d["L2_hit"] = round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4)
    tot = c.get("SQ_WAIT_ANY", 0) + c.get("SQ_WAIT_INST_ANY", 0) + c.get("SQ_ACTIVE_INST_ANY", 0)
    if tot:
        d["wave_time"] = dict(parked=round(c["SQ_WAIT_ANY"] / tot, 3), issue_stalled=round(c["SQ_WAIT_INST_ANY"] / tot, 3),
                              issuing=round(c["SQ_ACTIVE_INST_ANY"] / tot, 3))
    out[k] = d
line = json.loads(open(OUT + "/bench_line.json").read())
cfg = line["config"]
pts_per_frame = cfg["rays_per_gpu_per_step"] * cfg["samples_per_ray"]
ent = {}
import bench
for k, d in out.items():
    if k.endswith(line["roofline"]["kernel"]) and "hbm_read_GB_x2_corrected" in d:
        # the PMC passes ran 2 frames (1 warm-up + 1 step)
        bpp = (d["hbm_read_GB_x2_corrected"] + d["hbm_write_GB"]) * 1e9 / (2 * pts_per_frame)
        d["hbm_bytes_per_point"] = round(bpp, 1)
        name = line["roofline"]["kernel"]
        ent[name] = dict(hbm_bytes_per_point=round(bpp, 1), source=f"profiles/{os.path.basename(OUT)[5:]}_pmc_summary.json (FETCH_SIZE x2 per MI355X_MICROARCH.md + WRITE_SIZE, rocprofv3 PMC)",
                         source_digest=bench.kernel_source_digest(), workload=f"{cfg['frame']}x{cfg['samples_per_ray']}")
json.dump(out, open(OUT + "/pmc_summary.json", "w"), indent=1)
json.dump(ent, open(OUT + "/pmc_latest_entry.json", "w"), indent=1)
print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "counters"} for k, v in out.items()}, indent=1))
PY
cat $OUT/kernel_stats.md
cut -c1-400 $OUT/bench_line.json
