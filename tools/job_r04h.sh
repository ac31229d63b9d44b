#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04h; mkdir -p $O
timeout 1500 python -m pytest tests/test_train_gpu.py -x -q > $O/t_train.log 2>&1; echo "rc=$?" >> $O/t_train.log
tail -3 $O/t_train.log
timeout 900 python tools/time_train.py --objects 4 --rays 4096 --steps 3 > $O/time_train.txt 2>&1
timeout 600 python tools/time_train.py --objects 1 --rays 128 2048 --size 64x64 >> $O/time_train.txt 2>&1
grep -E "rays x" $O/time_train.txt
timeout 1200 tools/prof_train.sh 4096 1 > $O/prof_train.log 2>&1
cp gpurun_out/prof_train_1x4096/stats.md $O/train_1x4096_stats.md; cp gpurun_out/prof_train_1x4096/timeline.txt $O/train_1x4096_timeline.txt
timeout 1500 tools/pmc_train.sh 4096 1 > $O/pmc_train.log 2>&1
cp gpurun_out/pmc_train_1x4096/summary.md $O/train_1x4096_pmc.md
head -12 $O/train_1x4096_pmc.md
