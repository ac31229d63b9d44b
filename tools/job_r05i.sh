#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05i; mkdir -p $O
bash tools/job_r05g.sh prof 2>&1 | grep -E "==|clocks/wave|frontend|gather 0|sync|publish|gemm|store" | tee $O/phases.txt | cut -c1-110
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-26s rays/s %8.0f  ms/frame %8.2f  pre %7.3f ms/launch  frac %.4f  post total %.1f ms' % ('$1', d['value'], d['ms_per_step'], r.get('avg_launch_ms', -1), r['frac'], r['post_kernel_ms_total']))"; }
for i in 1 2; do
  DINER_F16_W8=1 timeout 600 python bench.py --gpus 1 --steps 5 --warmup 1 --precision f16 --no-modes --no-configs --cpu-rays 0 2>/dev/null | grep '^{' | tail -1 | line "800x600 f16 W8 base" | tee -a $O/ab.txt
  DINER_AMD_LIB=$PWD/diner_amd/libdiner_hip_r3.so timeout 600 python bench.py --gpus 1 --steps 5 --warmup 1 --precision f16 --no-modes --no-configs --cpu-rays 0 2>/dev/null | grep '^{' | tail -1 | line "800x600 f16 W8 ring0=3" | tee -a $O/ab.txt
  DINER_F16_W8=0 timeout 600 python bench.py --gpus 1 --steps 5 --warmup 1 --precision f16 --no-modes --no-configs --cpu-rays 0 2>/dev/null | grep '^{' | tail -1 | line "800x600 f16 4-wave" | tee -a $O/ab.txt
done
for v in 1 0; do
  DINER_F16_W8=$v timeout 900 python bench.py --gpus 1 --steps 2 --warmup 1 --precision f16 --width 1024 --height 1024 --samples 192 --facescape --no-modes --no-configs --cpu-rays 0 2>/dev/null | grep '^{' | tail -1 | line "1024^2 K192 f16 W8=$v" | tee -a $O/ab.txt
done
