#!/bin/bash
# round 5: f16x3 on the eight-wave kernel (DINER_F16X3_W8=1): parity tests of the f16x3 mode, then A/B on the headline
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05l; mkdir -p $O
DINER_F16X3_W8=1 timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_large_maps_gpu.py -m gpu -q -x -k "not fp32" > $O/t_x8.log 2>&1; echo "rc=$?" >> $O/t_x8.log
tail -5 $O/t_x8.log | cut -c1-250
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-26s rays/s %8.0f  ms/frame %8.2f  pre %7.3f ms/launch  frac %.4f  post total %.1f ms' % ('$1', d['value'], d['ms_per_step'], r.get('avg_launch_ms', -1), r['frac'], r['post_kernel_ms_total']))"; }
for i in 1 2; do
  for v in 1 0; do
  DINER_F16X3_W8=$v tools/power_sample.sh $O/pw_$v_$i.txt timeout 600 python bench.py --gpus 1 --steps 5 --warmup 1 --no-modes --no-configs --cpu-rays 0 2>/dev/null | grep '^{' | tail -1 | line "800x600 f16x3 W8=$v" | tee -a $O/ab.txt
  grep -E "power|sclk" $O/pw_$v_$i.txt | tr '\n' ' ' | tee -a $O/ab.txt; echo | tee -a $O/ab.txt
  done
done
