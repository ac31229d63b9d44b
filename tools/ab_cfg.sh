#!/bin/bash
# A/B of library variants on an arbitrary bench configuration (GPU box):
#   tools/ab_cfg.sh "<bench.py arguments>" name1 name2 ...     ("base" = the shipped library)
cd $GRAFT_REPO_ROOT
ARGS="$1"; shift
for n in "$@"; do
  if [ "$n" == "base" ]; then L=diner_amd/libdiner_hip.so; else L=diner_amd/libdiner_hip_$n.so; fi
  DINER_AMD_LIB=$PWD/$L python bench.py --steps 2 --warmup 1 --cpu-rays 0 --no-modes --no-configs $ARGS 2>/dev/null | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-10s rays/s %8.0f  pre %7.3f ms/launch  post total %7.2f ms  frac %.4f  whole %.4f' % ('$n', d['value'], r['avg_launch_ms'], r['post_kernel_ms_total'], r['frac'], r['whole_path']['frac']))" || echo "$n failed"
done
