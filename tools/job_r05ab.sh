#!/bin/bash
# round 5: fused training forward as the default (+ overflow repeat) -- whole training suite + timing with / without the read back
O=gpurun_out/r05ab; mkdir -p $O
timeout 1500 python -m pytest tests/test_train_gpu.py -x -q -m gpu 2>&1 | tail -8 | tee $O/pytest_train.log | cut -c1-250
for v in "1 1" "1 0" "0 1"; do
  set -- $v
  echo "== DINER_TRAIN_FUSED_FWD=$1 DINER_TRAIN_FUSED_CHECK=$2" | tee -a $O/time.txt
  DINER_TRAIN_FUSED_FWD=$1 DINER_TRAIN_FUSED_CHECK=$2 timeout 600 python tools/time_train.py --objects 4 --rays 4096 --steps 4 2>&1 | grep -E "rays x" | tee -a $O/time.txt | cut -c1-250
  DINER_TRAIN_FUSED_FWD=$1 DINER_TRAIN_FUSED_CHECK=$2 timeout 600 python tools/time_train.py --objects 4 --rays 128 --steps 20 2>&1 | grep -E "rays x" | tee -a $O/time.txt | cut -c1-250
done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o fused -- python $GRAFT_REPO_ROOT/tools/time_train.py --objects 1 --rays 4096 --steps 3 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
find $O/prof -name "*kernel_stats.csv" | head -2
f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -25 "$f" | cut -c1-200 | tee $O/kernel_stats_head.txt
find $O/prof -name "*.db" -delete
