#!/bin/bash
# round 5: fused training forward (DINER_TRAIN_FUSED_FWD=1) -- parity + A/B timing
O=gpurun_out/r05aa; mkdir -p $O
timeout 900 python -m pytest tests/test_train_gpu.py -x -q -m gpu -k "oracle_autograd" -s 2>&1 | tail -30 | tee $O/pytest_fused.log | cut -c1-250
for v in 0 1; do
  echo "== DINER_TRAIN_FUSED_FWD=$v" | tee -a $O/time.txt
  DINER_TRAIN_FUSED_FWD=$v timeout 600 python tools/time_train.py --objects 1 --rays 4096 --steps 5 2>&1 | grep -E "rays x" | tee -a $O/time.txt | cut -c1-250
  DINER_TRAIN_FUSED_FWD=$v timeout 600 python tools/time_train.py --objects 4 --rays 4096 --steps 3 2>&1 | grep -E "rays x" | tee -a $O/time.txt | cut -c1-250
done
cd /tmp && export TMPDIR=/tmp
DINER_TRAIN_FUSED_FWD=1 timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o fused -- python $GRAFT_REPO_ROOT/tools/time_train.py --objects 1 --rays 4096 --steps 3 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(ls $O/prof/*/fused_kernel_stats.csv $O/prof/fused_kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && head -25 "$f" | cut -c1-200 | tee $O/kernel_stats_head.txt
