#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04g; mkdir -p $O
timeout 1500 python -m pytest tests/test_train_gpu.py -x -q > $O/t_train.log 2>&1; echo "rc=$?" >> $O/t_train.log
tail -3 $O/t_train.log
for m in 1 0; do
  echo "== DINER_L512_T128=$m" >> $O/time_train.txt
  DINER_L512_T128=$m timeout 600 python tools/time_train.py --objects 1 --rays 4096 --steps 4 >> $O/time_train.txt 2>&1
done
grep -E "==|rays x" $O/time_train.txt
