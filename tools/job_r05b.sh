#!/bin/bash
# round 5: large-map oracle parity tests + the small-gradient training test + the training suite on the library with the amax fall-back fix
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05b; mkdir -p $O
timeout 1500 python -m pytest tests/test_large_maps_gpu.py -m gpu -q -s > $O/t_large.log 2>&1; echo "rc=$?" >> $O/t_large.log
grep -E "cfg3|cfg5|passed|failed|rc=|Error|assert" $O/t_large.log | cut -c1-330
timeout 1500 python -m pytest tests/test_train_gpu.py -m gpu -q -s > $O/t_train.log 2>&1; echo "rc=$?" >> $O/t_train.log
grep -E "upstream|passed|failed|rc=|Error" $O/t_train.log | cut -c1-250
