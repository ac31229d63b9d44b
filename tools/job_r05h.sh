#!/bin/bash
# round 5: 8-wave plain-fp16 kernel after the register fix: phases, fp16 tests, A/B vs the 4-wave kernel
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05h; mkdir -p $O
bash tools/job_r05g.sh prof 2>&1 | grep -E "==|clocks/wave|frontend|gather 0|sync|publish|gemm|store" | tee $O/phases.txt | cut -c1-110
timeout 1200 python -m pytest tests/test_hip_parity.py -m gpu -q -k "fp16 or f16 or tile_queues or replicated or cfg5" > $O/t_f16.log 2>&1; echo "rc=$?" >> $O/t_f16.log
tail -3 $O/t_f16.log
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-22s rays/s %8.0f  ms/frame %8.2f  pre %7.3f ms/launch  frac %.4f  post total %.1f ms' % ('$1', d['value'], d['ms_per_step'], r.get('avg_launch_ms', -1), r['frac'], r['post_kernel_ms_total']))"; }
for i in 1 2; do
  for v in 1 0; do
    DINER_F16_W8=$v timeout 600 python bench.py --gpus 1 --steps 5 --warmup 1 --precision f16 --no-modes --no-configs --cpu-rays 0 2>/dev/null | grep '^{' | tail -1 | line "800x600 f16 W8=$v" | tee -a $O/ab.txt
  done
done
for v in 1 0; do
  DINER_F16_W8=$v timeout 900 python bench.py --gpus 1 --steps 2 --warmup 1 --precision f16 --width 1024 --height 1024 --samples 192 --facescape --no-modes --no-configs --cpu-rays 0 2>/dev/null | grep '^{' | tail -1 | line "1024^2 K192 f16 W8=$v" | tee -a $O/ab.txt
done
