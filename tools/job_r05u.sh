#!/bin/bash
# round 5: eight-wave post kernel of the plain-fp16 mode: fp16 tests, A/B vs the four-wave post kernel (DINER_F16_POST_W8=0)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05u; mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_large_maps_gpu.py tests/test_boundary_gpu.py -m gpu -q -k "fp16 or f16 or tile_queues or replicated or cfg5 or overflow" > $O/t_f16.log 2>&1; echo "rc=$?" >> $O/t_f16.log
tail -3 $O/t_f16.log | cut -c1-200
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-30s rays/s %8.0f  ms/frame %8.2f  pre %7.3f ms/launch  frac %.4f  post total %.1f ms  whole %.4f' % ('$1', d['value'], d['ms_per_step'], r.get('avg_launch_ms', -1), r['frac'], r['post_kernel_ms_total'], r['whole_path']['frac']))"; }
for i in 1 2; do
  for v in 1 0; do
  DINER_F16_POST_W8=$v timeout 600 python bench.py --gpus 1 --steps 5 --warmup 1 --precision f16 --no-modes --no-configs --cpu-rays 0 2>/dev/null | grep '^{' | tail -1 | line "800x600 f16 post W8=$v" | tee -a $O/ab.txt
  done
done
CFG5="--facescape --width 1024 --height 1024 --samples 192"
for v in 1 0; do
DINER_F16_POST_W8=$v timeout 900 python bench.py --gpus 1 --steps 2 --warmup 1 --precision f16 $CFG5 --no-modes --no-configs --cpu-rays 0 2>/dev/null | grep '^{' | tail -1 | line "cfg4 f16 post W8=$v" | tee -a $O/ab.txt
done
