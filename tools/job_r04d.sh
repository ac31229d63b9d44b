#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04d; mkdir -p $O
timeout 1200 tools/prof_train.sh 4096 1 > $O/prof_train.log 2>&1
cp gpurun_out/prof_train_1x4096/stats.md $O/train_1x4096_stats.md; cp gpurun_out/prof_train_1x4096/timeline.txt $O/train_1x4096_timeline.txt
timeout 1500 python -m pytest tests/test_train_gpu.py -x -q -k "oracle_autograd" > $O/t_train.log 2>&1; echo "rc=$?" >> $O/t_train.log; grep -E "conditioned|passed|failed|rc=" $O/t_train.log | cut -c1-400
timeout 1500 python -m pytest tests/test_hip_parity.py -x -q -k "metric_sample or fp16 or f16" > $O/t_parity.log 2>&1; echo "rc=$?" >> $O/t_parity.log
grep -E "seed-to-seed|passed|failed|rc=" $O/t_parity.log | cut -c1-600
cat $O/train_1x4096_stats.md
