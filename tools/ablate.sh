#!/bin/bash
# Price the ingredients of the field kernels' stage loop: run the bench with ablated builds of the library.
# usage (on the GPU box): tools/ablate.sh name1 name2 ...   (libs diner_amd/libdiner_hip_<name>.so built beforehand)
cd $GRAFT_REPO_ROOT
for n in "$@"; do
  if [ "$n" == "base" ]; then L=diner_amd/libdiner_hip.so; else L=diner_amd/libdiner_hip_$n.so; fi
  DINER_AMD_LIB=$PWD/$L python bench.py --steps 1 --warmup 1 --cpu-rays 0 --no-configs --width 256 --height 256 2>&1 | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-14s rays/s %8.0f  pre %8.2f ms  post %7.2f ms  frac %.3f' % ('$n', d['value'], r['pre_kernel_ms_total'], r['post_kernel_ms_total'], r['frac']))" || echo "$n failed"
done
