#!/bin/bash
# round 6, job a: the refactored training path (persistent handle, batched step) -- training tests, step timing, lin512 phase timer
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_train_gpu.py -x -q --durations=0 > gpurun_out/r06a_train_tests.log 2>&1
echo "train tests rc=$?" >> gpurun_out/r06a_train_tests.log
tail -5 gpurun_out/r06a_train_tests.log
timeout 600 python tools/time_train.py --objects 4 --rays 4096 128 > gpurun_out/r06a_time_train.log 2>&1
DINER_TRAIN_BATCH=0 timeout 600 python tools/time_train.py --objects 4 --rays 4096 128 > gpurun_out/r06a_time_train_nobatch.log 2>&1
cat gpurun_out/r06a_time_train.log gpurun_out/r06a_time_train_nobatch.log
DINER_AMD_LIB=$PWD/diner_amd/libdiner_hip_l512prof.so timeout 600 python tools/prof_l512.py > gpurun_out/r06a_prof_l512.log 2>&1
cat gpurun_out/r06a_prof_l512.log
