#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04i; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/t_gpu_all.log 2>&1; echo "rc=$?" >> $O/t_gpu_all.log
tail -4 $O/t_gpu_all.log
timeout 900 tools/profile_round.sh r04i_800 > $O/prof_800.log 2>&1
CFG5="--facescape --width 1024 --height 1024 --samples 192"
timeout 900 tools/profile_round.sh r04i_cfg5_f16 $CFG5 --precision f16 > $O/prof_cfg5_f16.log 2>&1
timeout 900 tools/profile_round.sh r04i_cfg5_f16x3 $CFG5 --precision f16x3 > $O/prof_cfg5_f16x3.log 2>&1
timeout 900 tools/profile_round.sh r04i_800_f16 --precision f16 > $O/prof_800_f16.log 2>&1
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 2 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; echo "bench rc=$?" >> $O/bench_driver_cmd.err
timeout 900 python bench.py --gpus 2 --steps 3 --warmup 1 > $O/bench_2ranks.json 2> $O/bench_2ranks.err; echo "bench2 rc=$?" >> $O/bench_2ranks.err
cut -c1-300 $O/bench_driver_cmd.json
