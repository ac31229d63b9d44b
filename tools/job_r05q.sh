#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05q; mkdir -p $O
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-34s rays/s %8.0f  ms/frame %8.2f  pre %7.3f ms/launch  frac %.4f' % ('$1', d['value'], d['ms_per_step'], r.get('avg_launch_ms', -1), r['frac']))"; }
for p in 3 4 6 12 1 3; do
  DINER_QMAP_PASSES=$p timeout 900 python bench.py --gpus 1 --steps 2 --warmup 1 --width 1024 --height 1024 --samples 192 --facescape --no-modes --no-configs --cpu-rays 0 2>/dev/null | grep '^{' | tail -1 | line "cfg4 f16x3 passes=$p" | tee -a $O/ab.txt
done
for p in 1 3 6; do
  DINER_QMAP_PASSES=$p timeout 900 python bench.py --gpus 1 --steps 2 --warmup 1 --precision f16 --width 1024 --height 1024 --samples 192 --facescape --no-modes --no-configs --cpu-rays 0 2>/dev/null | grep '^{' | tail -1 | line "cfg4 f16 passes=$p" | tee -a $O/ab.txt
done
for p in 1 4 8 1; do
  DINER_QMAP_PASSES=$p timeout 900 python bench.py --gpus 1 --steps 4 --warmup 1 --no-modes --no-configs --cpu-rays 0 2>/dev/null | grep '^{' | tail -1 | line "800x600 f16x3 passes=$p" | tee -a $O/ab.txt
done
