#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04a
timeout 1500 python -m pytest tests/test_bench_selflaunch_gpu.py -x -q > gpurun_out/r04a/t_selflaunch.log 2>&1; echo "selflaunch rc=$?" >> gpurun_out/r04a/t_selflaunch.log
CFG5="--facescape --width 1024 --height 1024 --samples 192"
timeout 900 tools/profile_round.sh r04_cfg5_f16 $CFG5 --precision f16 > gpurun_out/r04a/prof_cfg5_f16.log 2>&1
timeout 900 tools/profile_round.sh r04_cfg5_f16x3 $CFG5 --precision f16x3 > gpurun_out/r04a/prof_cfg5_f16x3.log 2>&1
timeout 600 tools/prof_phases.sh $CFG5 --precision f16 > gpurun_out/r04a/phase_cfg5_f16.txt 2>&1
timeout 600 tools/prof_phases.sh $CFG5 --precision f16x3 > gpurun_out/r04a/phase_cfg5_f16x3.txt 2>&1
timeout 600 tools/prof_phases.sh --width 800 --height 600 --precision f16 > gpurun_out/r04a/phase_800_f16.txt 2>&1
timeout 600 tools/prof_phases.sh --width 800 --height 600 --precision f16x3 > gpurun_out/r04a/phase_800_f16x3.txt 2>&1
timeout 900 python bench.py --gpus 1 --steps 5 --warmup 1 > gpurun_out/r04a/bench_default.json 2> gpurun_out/r04a/bench_default.err; echo "bench rc=$?" >> gpurun_out/r04a/bench_default.err
tail -3 gpurun_out/r04a/t_selflaunch.log
