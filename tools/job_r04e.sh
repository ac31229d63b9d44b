#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04e; mkdir -p $O
timeout 1500 tools/pmc_train.sh 4096 1 > $O/pmc_train.log 2>&1
tail -25 $O/pmc_train.log
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -k "metric_sample" -s 2>&1 | grep -E "seed-to-seed|passed|failed" | cut -c1-700 > $O/t_s2s.log; cat $O/t_s2s.log
