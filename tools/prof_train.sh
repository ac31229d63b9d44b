#!/bin/bash
# rocprofv3 kernel trace of the training step at one batch size: per-kernel table + the sum of kernel time per step beside the wall time per
# step (the difference is launch gaps / host time).  usage: tools/prof_train.sh [rays per object = 4096] [objects = 4]
# (defaults: the step the shipped configs run, configs/train_dtu.yaml: 4 objects x 4096 rays x 40 samples).  Output under
# gpurun_out/prof_train_<objects>x<rays>/.
RAYS=${1:-4096}
OBJ=${2:-4}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_train_${OBJ}x$RAYS
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $GRAFT_REPO_ROOT/tools/time_train.py --objects $OBJ --rays $RAYS --steps 3 > $OUT/plain.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -o t -- python $GRAFT_REPO_ROOT/tools/time_train.py --objects $OBJ --rays $RAYS --steps 3 > $OUT/under_rocprof.txt 2>&1
python $GRAFT_REPO_ROOT/tools/summarize_rocprof.py $(find $OUT/raw -name "*kernel_stats.csv" | head -1) $OUT/stats.md "python tools/time_train.py --objects $OBJ --rays $RAYS --steps 3 (4 forward+backward steps)" > /dev/null 2>&1
OBJ=$OBJ python - $(find $OUT/raw -name '*kernel_trace.csv' | head -1) <<'PY' > $OUT/timeline.txt 2>&1
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last full step: from the last k_sample_depthguided to the end
import os
nobj = int(os.environ.get("OBJ", "1"))
idx = [i for i, r in enumerate(rows) if "k_sample_depthguided" in r["Kernel_Name"]]
a, b = idx[-2 * nobj], idx[-nobj]
step = rows[a:b]
t0 = int(step[0]["Start_Timestamp"]); t1 = int(rows[b]["Start_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in step)
print(f"one step: {len(step)} kernels, {(t1 - t0) / 1e6:.3f} ms from sampler to sampler, {busy / 1e6:.3f} ms inside kernels")
prev = t0
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  gap {(s - prev) / 1e3:6.1f}  dur {(e - s) / 1e3:7.1f}  {r['Kernel_Name'][:90]}")
    prev = e
PY
rm -rf $OUT/raw
cat $OUT/plain.txt $OUT/under_rocprof.txt | grep rays; head -3 $OUT/timeline.txt
