#!/bin/bash
# HBM traffic of the training step's kernels (GPU box): rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE, one per pass as MI355X_MICROARCH.md
# prescribes; FETCH_SIZE doubled for wide coalesced reads) + MFMA busy, per kernel, beside the kernel time of the same run.
# usage: tools/pmc_train.sh [rays per object = 4096] [objects = 1]  -> gpurun_out/pmc_train_<objects>x<rays>/summary.{json,md}
RAYS=${1:-4096}; OBJ=${2:-1}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_train_${OBJ}x$RAYS; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/tools/time_train.py --objects $OBJ --rays $RAYS --steps 1"
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  n=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_$n -o pmc -- $CMD > $OUT/pmc_$n.log 2>&1
done
python - <<PY
import csv, glob, collections, json
OUT = "$OUT"
acc = collections.defaultdict(lambda: collections.defaultdict(float)); dur = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob(OUT + "/pmc_*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]] += float(r["Counter_Value"])
f = glob.glob(OUT + "/pmc_FETCH_SIZE/*kernel_trace.csv")[0]
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0]
    dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9; n[k] += 1
rows = []
for k in acc:
    c = acc[k]
    rd, wr = c.get("FETCH_SIZE", 0) * 2 * 1024 / 1e9, c.get("WRITE_SIZE", 0) * 1024 / 1e9
    g = c.get("GRBM_GUI_ACTIVE", 0) / 8.0
    rows.append(dict(kernel=k, launches=n[k], time_ms=round(dur[k] * 1e3, 3), hbm_read_GB=round(rd, 3), hbm_write_GB=round(wr, 3),
                     TBps=round((rd + wr) / dur[k] / 1e3, 3) if dur[k] else None,
                     MfmaUtil=round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (g * 1024), 4) if g else None))
rows.sort(key=lambda r: -r["time_ms"])
json.dump(rows, open(OUT + "/summary.json", "w"), indent=1)
tot_t = sum(r["time_ms"] for r in rows); tot_b = sum(r["hbm_read_GB"] + r["hbm_write_GB"] for r in rows)
with open(OUT + "/summary.md", "w") as md:
    md.write(f"command: $CMD (2 steps: warm-up + 1)\n\n| kernel | launches | ms | HBM read GB (x2) | write GB | TB/s | MfmaUtil |\n|---|---|---|---|---|---|---|\n")
    for r in rows[:14]:
        md.write(f"| {r['kernel'][:60]} | {r['launches']} | {r['time_ms']} | {r['hbm_read_GB']} | {r['hbm_write_GB']} | {r['TBps']} | {r['MfmaUtil']} |\n")
    md.write(f"\nall kernels: {tot_t:.1f} ms, {tot_b:.1f} GB = {tot_b / tot_t:.2f} TB/s\n")
print(open(OUT + "/summary.md").read())
PY
