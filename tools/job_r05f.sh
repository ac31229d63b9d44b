#!/bin/bash
# round 5: phase timer of the plain-fp16 per-view kernels (8-wave vs 4-wave), 800x600
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05f; mkdir -p $O
for v in 1 0; do
  echo "== DINER_F16_W8=$v" | tee -a $O/phases.txt
  DINER_F16_W8=$v tools/prof_phases.sh --precision f16 2>&1 | grep -v "prof post" | tee -a $O/phases.txt
done
