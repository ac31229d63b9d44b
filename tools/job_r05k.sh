#!/bin/bash
# round 5: generic-shape slow path tests + the boundary / parity suites on ABI v5
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05k; mkdir -p $O
timeout 900 python -m pytest tests/test_generic_gpu.py -m gpu -q -s > $O/t_generic.log 2>&1; echo "rc=$?" >> $O/t_generic.log
grep -E "ResnetFC case|generic PixelNeRF|passed|failed|rc=|Error|error" $O/t_generic.log | cut -c1-300
timeout 1500 python -m pytest tests/test_boundary_gpu.py tests/test_hip_parity.py -m gpu -q -x > $O/t_parity.log 2>&1; echo "rc=$?" >> $O/t_parity.log
tail -4 $O/t_parity.log | cut -c1-200
