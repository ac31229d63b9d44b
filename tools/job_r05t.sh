#!/bin/bash
# round 5: 8-wave plain-fp16 kernel with per-wave publish flags instead of a barrier per layer: fp16 tests, phases, A/B vs the barrier build (nf)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05t; mkdir -p $O
timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_large_maps_gpu.py -m gpu -q -k "fp16 or f16 or tile_queues or replicated or cfg5" > $O/t_f16.log 2>&1; echo "rc=$?" >> $O/t_f16.log
tail -3 $O/t_f16.log | cut -c1-200
bash tools/job_r05g.sh prof 2>&1 | grep -E "==|clocks/wave|frontend|gather 0|sync|publish|gemm|store" | tee $O/phases.txt | cut -c1-110
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-30s rays/s %8.0f  ms/frame %8.2f  pre %7.3f ms/launch  frac %.4f' % ('$1', d['value'], d['ms_per_step'], r.get('avg_launch_ms', -1), r['frac']))"; }
for i in 1 2; do
  timeout 600 python bench.py --gpus 1 --steps 5 --warmup 1 --precision f16 --no-modes --no-configs --cpu-rays 0 2>/dev/null | grep '^{' | tail -1 | line "800x600 f16 flags" | tee -a $O/ab.txt
  DINER_AMD_LIB=$PWD/diner_amd/libdiner_hip_nf.so timeout 600 python bench.py --gpus 1 --steps 5 --warmup 1 --precision f16 --no-modes --no-configs --cpu-rays 0 2>/dev/null | grep '^{' | tail -1 | line "800x600 f16 barriers" | tee -a $O/ab.txt
done
CFG5="--facescape --width 1024 --height 1024 --samples 192"
timeout 900 python bench.py --gpus 1 --steps 2 --warmup 1 --precision f16 $CFG5 --no-modes --no-configs --cpu-rays 0 2>/dev/null | grep '^{' | tail -1 | line "cfg4 f16 flags" | tee -a $O/ab.txt
DINER_AMD_LIB=$PWD/diner_amd/libdiner_hip_nf.so timeout 900 python bench.py --gpus 1 --steps 2 --warmup 1 --precision f16 $CFG5 --no-modes --no-configs --cpu-rays 0 2>/dev/null | grep '^{' | tail -1 | line "cfg4 f16 barriers" | tee -a $O/ab.txt
