#!/bin/bash
# round 5: fused training forward with the gated layer-wise repeat on the device (no flag read back)
O=gpurun_out/r05ae; mkdir -p $O
timeout 1500 python -m pytest tests/test_train_gpu.py -x -q -m gpu -s 2>&1 | grep -E "training path|conditioned|Frobenius|beyond|passed|failed|Error|error" | tee $O/pytest_train.log | cut -c1-250
for v in 1 0; do
  echo "== DINER_TRAIN_FUSED_FWD=$v" | tee -a $O/time.txt
  DINER_TRAIN_FUSED_FWD=$v timeout 600 python tools/time_train.py --objects 4 --rays 4096 --steps 4 2>&1 | grep -E "rays x" | tee -a $O/time.txt | cut -c1-250
  DINER_TRAIN_FUSED_FWD=$v timeout 600 python tools/time_train.py --objects 1 --rays 4096 --steps 5 2>&1 | grep -E "rays x" | tee -a $O/time.txt | cut -c1-250
done
timeout 600 python tools/time_train.py --objects 4 --rays 128 --steps 20 2>&1 | grep -E "rays x" | tee -a $O/time.txt | cut -c1-250
