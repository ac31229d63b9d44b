#!/bin/bash
# round 5: fused training forward with the in-library f16x3 projection of the latent map; parameters bumped every step in the timing harness
O=gpurun_out/r05ad; mkdir -p $O
timeout 1500 python -m pytest tests/test_train_gpu.py -x -q -m gpu -s 2>&1 | grep -E "training path|conditioned|Frobenius|beyond|passed|failed|Error|error" | tee $O/pytest_train.log | cut -c1-250
for v in "1 1" "1 0" "0 1"; do
  set -- $v
  echo "== DINER_TRAIN_FUSED_FWD=$1 DINER_TRAIN_FUSED_CHECK=$2" | tee -a $O/time.txt
  DINER_TRAIN_FUSED_FWD=$1 DINER_TRAIN_FUSED_CHECK=$2 timeout 600 python tools/time_train.py --objects 4 --rays 4096 --steps 4 2>&1 | grep -E "rays x" | tee -a $O/time.txt | cut -c1-250
  DINER_TRAIN_FUSED_FWD=$1 DINER_TRAIN_FUSED_CHECK=$2 timeout 600 python tools/time_train.py --objects 1 --rays 4096 --steps 5 2>&1 | grep -E "rays x" | tee -a $O/time.txt | cut -c1-250
done
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o fused -- python $R/tools/time_train.py --objects 1 --rays 4096 --steps 4 > $R/$O/prof.log 2>&1
cd $R
f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
python tools/summarize_rocprof.py "$f" $O/r05_train_fused_kernel_stats.md "python tools/time_train.py --objects 1 --rays 4096 --steps 4 (5 steps: 1 warm-up + 4 timed)" 30 > /dev/null 2>&1
head -22 $O/r05_train_fused_kernel_stats.md | cut -c1-200
find $O/prof -name "*.db" -delete; find $O/prof -name "*trace.csv" -delete
