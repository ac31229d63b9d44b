#!/bin/bash
# Wave-time split, LDS bank conflicts and vector-memory path busy of the training step's kernels (separate rocprofv3 --pmc passes).
# usage: tools/pmc_train_waves.sh [rays = 4096] [objects = 1] -> gpurun_out/pmc_train_waves/summary.txt
RAYS=${1:-4096}; OBJ=${2:-1}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_train_waves; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/tools/time_train.py --objects $OBJ --rays $RAYS --steps 1"
i=0
for grp in "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TA_TA_BUSY_sum GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -o pmc -- $CMD > $OUT/p$i.log 2>&1
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("$OUT/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "512" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
with open("$OUT/summary.txt", "w") as out:
    out.write("command: $CMD\n")
    for k, a in acc.items():
        out.write(k + "\n")
        for c, v in sorted(a.items()): out.write(f"   {c:36s} {v:.4e}\n")
        tot = a.get("SQ_WAIT_ANY", 0) + a.get("SQ_WAIT_INST_ANY", 0) + a.get("SQ_ACTIVE_INST_ANY", 0)
        if tot: out.write(f"   wave time: parked {a['SQ_WAIT_ANY']/tot:.3f} issue-stalled {a['SQ_WAIT_INST_ANY']/tot:.3f} issuing {a['SQ_ACTIVE_INST_ANY']/tot:.3f}\n")
        if a.get("GRBM_GUI_ACTIVE"):
            cu = a["GRBM_GUI_ACTIVE"] / 8.0 * 256
            out.write(f"   TA busy {a.get('TA_TA_BUSY_sum', 0)/cu:.3f}   MfmaUtil {a.get('SQ_VALU_MFMA_BUSY_CYCLES', 0)/(a['GRBM_GUI_ACTIVE']/8.0*1024):.3f}\n")
        if a.get("SQ_LDS_IDX_ACTIVE"): out.write(f"   LDS bank-conflict cycles / LDS active cycles {a.get('SQ_LDS_BANK_CONFLICT', 0)/a['SQ_LDS_IDX_ACTIVE']:.3f}\n")
        if a.get("TCP_TCC_READ_REQ_sum"): out.write(f"   mean L1->L2 read latency {a['TCP_TCC_READ_REQ_LATENCY_sum']/a['TCP_TCC_READ_REQ_sum']:.0f} clocks\n")
print(open("$OUT/summary.txt").read())
PY
