#!/bin/bash
# round 6, job b: the 128-row f16x3 shape's epilogue through LDS -- training tests, same-box A/B against the direct epilogue, phase timer
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_train_gpu.py -x -q --durations=0 > gpurun_out/r06b_train_tests.log 2>&1
echo "train tests rc=$?" >> gpurun_out/r06b_train_tests.log
tail -4 gpurun_out/r06b_train_tests.log
for rep in 1 2; do
  for lib in "" oldepi; do
    if [ -n "$lib" ]; then export DINER_AMD_LIB=$PWD/diner_amd/libdiner_hip_$lib.so; else unset DINER_AMD_LIB; fi
    echo "== lib ${lib:-default} (rep $rep)" >> gpurun_out/r06b_time_train.log
    timeout 600 python tools/time_train.py --objects 4 --rays 4096 >> gpurun_out/r06b_time_train.log 2>&1
  done
done
unset DINER_AMD_LIB
grep -v amdgpu.ids gpurun_out/r06b_time_train.log
DINER_AMD_LIB=$PWD/diner_amd/libdiner_hip_l512prof.so timeout 600 python tools/prof_l512.py > gpurun_out/r06b_prof_l512.log 2>&1
grep -v amdgpu.ids gpurun_out/r06b_prof_l512.log
