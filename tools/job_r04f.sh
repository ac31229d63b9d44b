#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04f; mkdir -p $O
timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_boundary_gpu.py -x -q > $O/t_parity.log 2>&1; echo "rc=$?" >> $O/t_parity.log
tail -4 $O/t_parity.log
CFG5="--facescape --width 1024 --height 1024 --samples 192"
echo "== cfg5 f16" >> $O/ab.txt;    tools/ab_cfg.sh "$CFG5 --precision f16" base hd3 hd3r6 hd1 >> $O/ab.txt 2>&1
echo "== 800x600 f16" >> $O/ab.txt; tools/ab_cfg.sh "--precision f16" base hd3 hd3r6 hd1 >> $O/ab.txt 2>&1
cat $O/ab.txt
