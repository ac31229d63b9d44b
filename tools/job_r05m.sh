#!/bin/bash
# phase timer of the f16x3 per-view kernels: eight-wave (DINER_F16X3_W8=1) vs four-wave, 256x256
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05m; mkdir -p $O
for v in 1 0; do
  echo "== DINER_F16X3_W8=$v" | tee -a $O/phases.txt
  DINER_F16X3_W8=$v DINER_AMD_LIB=$PWD/diner_amd/libdiner_hip_prof.so python bench.py --steps 1 --warmup 1 --cpu-rays 0 --no-modes --no-configs --width 256 --height 256 2>&1 | grep "h3n prof\]" | tail -16 | tee -a $O/phases.txt
done
