#!/bin/bash
# round 5: k_train_fwd_pre with its workgroups started in four phase groups (DINER_TRAIN_SAVE_STAGGER = s_sleep(127) units per phase step)
O=gpurun_out/r05ai; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in 0 5 10 20 0 10; do
  DINER_TRAIN_SAVE_STAGGER=$v timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$v -o t -- python $R/tools/time_train.py --objects 1 --rays 4096 --steps 4 > $R/$O/prof_$v.log 2>&1
  f=$(find $R/$O/prof_$v -name "*kernel_stats.csv" | head -1)
  echo "== stagger $v" | tee -a $R/$O/summary.txt
  grep "rays x" $R/$O/prof_$v.log | cut -c1-120 | tee -a $R/$O/summary.txt
  grep -E "k_train_fwd_pre" $f | cut -c1-160 | tee -a $R/$O/summary.txt
  rm -rf $R/$O/prof_$v
done
