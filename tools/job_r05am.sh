#!/bin/bash
# round 5: data gradient and weight gradient of a layer as two launches (DINER_TRAIN_BWD_SPLIT=1): their separate durations, and what the split costs
O=gpurun_out/r05am; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  DINER_TRAIN_BWD_SPLIT=$v timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$v -o t -- python $R/tools/time_train.py --objects 1 --rays 4096 --steps 3 > $R/$O/prof_$v.log 2>&1
  echo "== split $v" | tee -a $R/$O/summary.txt
  grep "rays x" $R/$O/prof_$v.log | cut -c1-120 | tee -a $R/$O/summary.txt
  f=$(find $R/$O/prof_$v -name "*kernel_stats.csv" | head -1)
  grep -E "k_run512_f16x3" $f | cut -c1-160 | tee -a $R/$O/summary.txt
done
python - <<PY | tee -a $R/$O/summary.txt
import csv, glob
f = glob.glob("$R/$O/prof_1/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_run512_f16x3" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-26:]          # the last step: 13 layers x (data gradient, weight gradient); lin_z layers have both too
for i, r in enumerate(last):
    print(i, "grid", r.get("Grid_Size", r.get("Grid_Size_X", "?")), "ms %.3f" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
PY
find $R/$O -name "*.db" -delete; find $R/$O -name "*trace.csv" -delete
