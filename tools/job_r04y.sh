#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04y; mkdir -p $O
CFG5="--facescape --width 1024 --height 1024 --samples 192"
timeout 900 tools/profile_round.sh r04z_800 > $O/prof_800.log 2>&1
timeout 900 tools/profile_round.sh r04z_800_f16 --precision f16 > $O/prof_800_f16.log 2>&1
timeout 900 tools/profile_round.sh r04z_cfg5_f16 $CFG5 --precision f16 > $O/prof_cfg5_f16.log 2>&1
timeout 900 tools/profile_round.sh r04z_cfg5_f16x3 $CFG5 --precision f16x3 > $O/prof_cfg5_f16x3.log 2>&1
tail -5 $O/prof_800.log | cut -c1-300
