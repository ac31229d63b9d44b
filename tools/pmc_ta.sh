#!/bin/bash
# Vector-memory path counters of the field kernels (TA / TCP), separate rocprofv3 --pmc passes.
# usage: tools/pmc_ta.sh <tag> [bench.py arguments]   (default workload: 256 x 256)  -> gpurun_out/pmc_ta_<tag>/summary.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${1:-x}; shift
ARGS="${@:---width 256 --height 256}"
OUT=gpurun_out/pmc_ta_$TAG; mkdir -p $OUT
CMD="python bench.py --steps 1 --warmup 1 --cpu-rays 0 --no-modes --no-configs $ARGS"
i=0
for grp in "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE TA_FLAT_READ_WAVEFRONTS_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -o pmc -- $CMD > $OUT/p$i.log 2>&1
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("$OUT/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "k_field" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
with open("$OUT/summary.txt", "w") as out:
    out.write("command: $CMD\n")
    for k in acc:
        for c, v in sorted(acc[k].items()):
            out.write(f"{k:36s} {c:40s} total {v:.4e} launches {n[(k,c)]}\n")
        a = acc[k]
        if a.get("GRBM_GUI_ACTIVE") and a.get("TA_TA_BUSY_sum"):
            cu_cycles = a["GRBM_GUI_ACTIVE"] / 8.0 * 256
            out.write(f"{k:36s} TA busy = TA_TA_BUSY_sum / (256 CUs x kernel clocks) = {a['TA_TA_BUSY_sum'] / cu_cycles:.3f}\n")
        if a.get("TCP_TCC_READ_REQ_sum") and a.get("TCP_TCC_READ_REQ_LATENCY_sum"):
            out.write(f"{k:36s} mean L1->L2 read latency = {a['TCP_TCC_READ_REQ_LATENCY_sum'] / a['TCP_TCC_READ_REQ_sum']:.0f} clocks\n")
print(open("$OUT/summary.txt").read())
PY
