#!/bin/bash
# round 5, diagnostic for the next round (the TCP_*_LATENCY counters hung the profiler for 25 min at the first try: dropped, every pass under `timeout 60`; result: profiles/r05_train_fwd_pre_store_counters.txt): counters of the per-view training kernel with and without its stores (DINER_TRAIN_NOSAVE=1)
O=gpurun_out/r05aq; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/time_train.py --objects 1 --rays 4096 --steps 1"
for v in 0 1; do
  for grp in "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_HIT_sum TCC_MISS_sum TCC_WRITEBACK_sum" "TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum"; do
    n=$(echo $grp | cut -d' ' -f1)
    DINER_TRAIN_NOSAVE=$v timeout 60 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/$O/p${v}_$n -o pmc -- $CMD > $R/$O/p${v}_$n.log 2>&1
  done
done
cd $R
python - <<PY | tee $O/summary.txt
import csv, glob, collections
for v in (0, 1):
    acc = collections.defaultdict(float)
    for f in glob.glob("$O/p%d_*/**/*counter_collection.csv" % v, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "k_train_fwd_pre" in k or "k_field_pre_h3n" in k: acc[r["Counter_Name"]] += float(r["Counter_Value"])
    print("== per-view kernel,", "with its stores" if v == 0 else "DINER_TRAIN_NOSAVE=1 (the inference kernel in its place)", "-- 2 launches")
    for c, x in sorted(acc.items()): print(f"   {c:36s} {x:.4e}")
PY
rm -rf $O/p0_* $O/p1_*
