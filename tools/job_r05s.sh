#!/bin/bash
# round 5 profile pass: rocprofv3 kernel stats + PMC of the four measured workloads (headline f16x3, 800x600 f16, configs[4] f16 / f16x3)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05s; mkdir -p $O
CFG5="--facescape --width 1024 --height 1024 --samples 192"
timeout 900 tools/profile_round.sh r05s_800 > $O/prof_800.log 2>&1
timeout 900 tools/profile_round.sh r05s_800_f16 --precision f16 > $O/prof_800_f16.log 2>&1
timeout 900 tools/profile_round.sh r05s_cfg5_f16 $CFG5 --precision f16 > $O/prof_cfg5_f16.log 2>&1
timeout 900 tools/profile_round.sh r05s_cfg5_f16x3 $CFG5 --precision f16x3 > $O/prof_cfg5_f16x3.log 2>&1
for t in 800 800_f16 cfg5_f16 cfg5_f16x3; do echo "== $t"; tail -3 $O/prof_$t.log | cut -c1-300; done
