#!/bin/bash
# round 5: training forward products with two workgroups per CU (DINER_L512_W2=1) -- correctness of the 512-layer test, then step timing A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05o; mkdir -p $O
DINER_L512_W2=1 timeout 600 python -m pytest tests/test_train_gpu.py -m gpu -q -k "linear512 or oracle_autograd" 2>&1 | tail -3 | tee $O/t_w2.log
for v in 0 1 0 1; do
  echo "== DINER_L512_W2=$v" | tee -a $O/time.txt
  DINER_L512_W2=$v timeout 600 python tools/time_train.py --objects 1 --rays 4096 --steps 5 2>&1 | grep -E "rays x|ms" | tail -3 | tee -a $O/time.txt | cut -c1-200
done
