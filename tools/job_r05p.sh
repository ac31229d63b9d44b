#!/bin/bash
# round 5: tile-queue dealing variants (DINER_QMAP_PASSES / _RMUL) on BASELINE configs[4] in f16x3 and on the headline
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05p; mkdir -p $O
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-34s rays/s %8.0f  ms/frame %8.2f  pre %7.3f ms/launch  frac %.4f' % ('$1', d['value'], d['ms_per_step'], r.get('avg_launch_ms', -1), r['frac']))"; }
for v in "1 1" "2 1" "3 1" "1 4" "2 2" "1 1"; do
  set -- $v
  DINER_QMAP_PASSES=$1 DINER_QMAP_RMUL=$2 timeout 900 python bench.py --gpus 1 --steps 2 --warmup 1 --width 1024 --height 1024 --samples 192 --facescape --no-modes --no-configs --cpu-rays 0 2>/dev/null | grep '^{' | tail -1 | line "cfg4 f16x3 passes=$1 rmul=$2" | tee -a $O/ab.txt
done
for v in "1 1" "2 1" "1 2" "1 1" "2 1"; do
  set -- $v
  DINER_QMAP_PASSES=$1 DINER_QMAP_RMUL=$2 timeout 900 python bench.py --gpus 1 --steps 4 --warmup 1 --no-modes --no-configs --cpu-rays 0 2>/dev/null | grep '^{' | tail -1 | line "800x600 f16x3 passes=$1 rmul=$2" | tee -a $O/ab.txt
done
