#!/bin/bash
# round 5: kernel statistics of the training step with the fused forward (one object, 4096 rays x 40 samples)
O=gpurun_out/r05ac; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o fused -- python $R/tools/time_train.py --objects 1 --rays 4096 --steps 4 > $R/$O/prof.log 2>&1
cd $R
f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
python tools/summarize_rocprof.py "$f" profiles/r05_train_fused_kernel_stats.md "python tools/time_train.py --objects 1 --rays 4096 --steps 4 (7 launches of the step: 1 build + 2 warm-up + 4 timed)" 30 2>&1 | tail -3
cp profiles/r05_train_fused_kernel_stats.md $O/
head -40 $O/r05_train_fused_kernel_stats.md | cut -c1-220
find $O/prof -name "*.db" -delete; find $O/prof -name "*trace.csv" -delete
