#!/bin/bash
# round 5: relu decisions as bits for the backward's data gradients -- tests + A/B timing
O=gpurun_out/r05ag; mkdir -p $O
timeout 1500 python -m pytest tests/test_train_gpu.py -x -q -m gpu -s 2>&1 | grep -E "training path|conditioned|Frobenius|beyond|passed|failed|Error|error|assert" | tee $O/pytest_train.log | cut -c1-250
for v in 1 0 1 0; do
  echo "== DINER_TRAIN_MASKBITS=$v" | tee -a $O/time.txt
  DINER_TRAIN_MASKBITS=$v timeout 600 python tools/time_train.py --objects 1 --rays 4096 --steps 6 2>&1 | grep -E "rays x" | tee -a $O/time.txt | cut -c1-250
done
DINER_TRAIN_MASKBITS=1 timeout 600 python tools/time_train.py --objects 4 --rays 4096 --steps 4 2>&1 | grep -E "rays x" | tee -a $O/time.txt | cut -c1-250
DINER_TRAIN_MASKBITS=0 timeout 600 python tools/time_train.py --objects 4 --rays 4096 --steps 4 2>&1 | grep -E "rays x" | tee -a $O/time.txt | cut -c1-250
