#!/bin/bash
# round 5: the 8-wave plain-fp16 per-view kernel: fp16-mode tests, then A/B against the 4-wave kernel (DINER_F16_W8=0) at 800x600 and 1024^2 K=192
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05e; mkdir -p $O
timeout 1200 python -m pytest tests/test_hip_parity.py -m gpu -q -s -k "fp16 or f16 or tile_queues or replicated or cfg5" > $O/t_f16.log 2>&1; echo "rc=$?" >> $O/t_f16.log
grep -E "passed|failed|rc=|^FAILED|Error|dB" $O/t_f16.log | cut -c1-250 | tail -20
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-22s rays/s %8.0f  ms/frame %8.2f  pre %7.3f ms/launch  frac %.4f  post total %.1f ms' % ('$1', d['value'], d['ms_per_step'], r.get('avg_launch_ms', -1), r['frac'], r['post_kernel_ms_total']))"; }
for i in 1 2; do
  for v in 1 0; do
    DINER_F16_W8=$v timeout 600 python bench.py --gpus 1 --steps 5 --warmup 1 --precision f16 --no-modes --no-configs --cpu-rays 0 2>/dev/null | grep '^{' | tail -1 | line "800x600 f16 W8=$v" | tee -a $O/ab.txt
  done
done
for v in 1 0; do
  DINER_F16_W8=$v timeout 900 python bench.py --gpus 1 --steps 2 --warmup 1 --precision f16 --width 1024 --height 1024 --samples 192 --facescape --no-modes --no-configs --cpu-rays 0 2>/dev/null | grep '^{' | tail -1 | line "1024^2 K192 f16 W8=$v" | tee -a $O/ab.txt
done
