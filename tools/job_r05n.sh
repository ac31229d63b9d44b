#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05n; mkdir -p $O
for n in "$@"; do
  echo "== $n" | tee -a $O/phases.txt
  DINER_F16X3_W8=1 DINER_AMD_LIB=$PWD/diner_amd/libdiner_hip_$n.so python bench.py --steps 1 --warmup 1 --cpu-rays 0 --no-modes --no-configs --width 256 --height 256 2>&1 | grep "h3n prof\]" | tail -16 | tee -a $O/phases.txt
done
