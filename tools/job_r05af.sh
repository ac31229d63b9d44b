#!/bin/bash
# round 5: P = 20480 oracle-compared training test + PMC (HBM bytes, MfmaUtil) of the training step with the fused forward
O=gpurun_out/r05af; mkdir -p $O
timeout 1500 python -m pytest tests/test_train_gpu.py -x -q -m gpu -s -k "oracle_autograd" 2>&1 | grep -E "training path|conditioned|Frobenius|passed|failed|Error|error|assert" | tee $O/pytest_train.log | cut -c1-250
timeout 1500 tools/pmc_train.sh 4096 1 > $O/pmc_train.log 2>&1
tail -22 $O/pmc_train.log | cut -c1-200
cp gpurun_out/pmc_train_1x4096/summary.md $O/pmc_train_summary.md; cp gpurun_out/pmc_train_1x4096/summary.json $O/pmc_train_summary.json
rm -rf gpurun_out/pmc_train_1x4096/pmc_*
