#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c; mkdir -p $O
timeout 1500 python -m pytest tests/test_train_gpu.py -x -q > $O/t_train.log 2>&1; echo "rc=$?" >> $O/t_train.log
tail -5 $O/t_train.log
for m in 1 0; do
  echo "== DINER_TRAIN_BWD_F16X3=$m" >> $O/time_train.txt
  DINER_TRAIN_BWD_F16X3=$m timeout 600 python tools/time_train.py --objects 1 --rays 128 2048 --size 64x64 >> $O/time_train.txt 2>&1
  DINER_TRAIN_BWD_F16X3=$m timeout 900 python tools/time_train.py --objects 4 --rays 4096 --steps 3 >> $O/time_train.txt 2>&1
done
cat $O/time_train.txt | grep -E "==|rays x"
