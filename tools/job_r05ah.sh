#!/bin/bash
# round 5: what the stores of k_train_fwd_pre cost (no stores / non-temporal stores / plain), one object x 4096 rays
O=gpurun_out/r05ah; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in "plain 0 0" "nosave 1 0" "nt 0 1"; do
  set -- $v
  DINER_TRAIN_NOSAVE=$2 DINER_TRAIN_SAVE_NT=$3 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$1 -o t -- python $R/tools/time_train.py --objects 1 --rays 4096 --steps 4 > $R/$O/prof_$1.log 2>&1
  f=$(find $R/$O/prof_$1 -name "*kernel_stats.csv" | head -1)
  echo "== $1" | tee -a $R/$O/summary.txt
  grep "rays x" $R/$O/prof_$1.log | cut -c1-120 | tee -a $R/$O/summary.txt
  grep -E "k_train_fwd_pre|k_field_pre_h3n|k_train_fwd_post|k_run512_f16x3" $f | cut -c1-160 | tee -a $R/$O/summary.txt
  find $R/$O/prof_$1 -name "*.db" -delete; find $R/$O/prof_$1 -name "*trace.csv" -delete
done
