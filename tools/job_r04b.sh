#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04b; mkdir -p $O
CFG5="--facescape --width 1024 --height 1024 --samples 192"
for n in base r4g3 r6g4 r8g4 r6g2; do
  if [ "$n" == "base" ]; then L=diner_amd/libdiner_hip.so; else L=diner_amd/libdiner_hip_$n.so; fi
  echo "hash $n: $(DINER_AMD_LIB=$PWD/$L python tools/field_hash.py 2>/dev/null | tr '\n' ' ')" >> $O/ab.txt
done
echo "== cfg5 f16" >> $O/ab.txt;    tools/ab_cfg.sh "$CFG5 --precision f16" base r4g3 r6g4 r8g4 r6g2 >> $O/ab.txt 2>&1
echo "== cfg5 f16x3" >> $O/ab.txt;  tools/ab_cfg.sh "$CFG5 --precision f16x3" base >> $O/ab.txt 2>&1
echo "== 800x600 f16" >> $O/ab.txt; tools/ab_cfg.sh "--precision f16" base r4g3 r6g4 r8g4 r6g2 >> $O/ab.txt 2>&1
echo "== 800x600 f16x3" >> $O/ab.txt; tools/ab_cfg.sh "" base >> $O/ab.txt 2>&1
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_boundary_gpu.py -x -q > $O/t_parity.log 2>&1; echo "rc=$?" >> $O/t_parity.log
cat $O/ab.txt; tail -3 $O/t_parity.log
