#!/bin/bash
# round 5: phase timer of variants of the 8-wave plain-fp16 per-view kernel (800x600 would take longer: the default 256x256)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05g; mkdir -p $O
for n in "$@"; do
  echo "== $n" | tee -a $O/phases.txt
  DINER_AMD_LIB=$PWD/diner_amd/libdiner_hip_$n.so python bench.py --steps 1 --warmup 1 --cpu-rays 0 --no-modes --no-configs --width 256 --height 256 --precision f16 2>&1 | grep "h3n prof\]" | tail -16 | tee -a $O/phases.txt
done
