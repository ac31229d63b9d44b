#!/bin/bash
# round 5: whole GPU suite on the library with the fused unnormalize (ATen's vectorised CPU kernel contracts it) + the amax fix + bench self-launch fields
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05c; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q -s > $O/t_gpu_all.log 2>&1; echo "rc=$?" >> $O/t_gpu_all.log
grep -E "bilinear|passed|failed|rc=|^FAILED|Error" $O/t_gpu_all.log | cut -c1-300 | tail -30
