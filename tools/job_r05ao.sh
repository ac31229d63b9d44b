#!/bin/bash
# round 5: the data gradient on eight waves (k_dgrad512_w8, DINER_DGRAD_W8=1) -- parity + same-box A/B
O=gpurun_out/r05ao; mkdir -p $O
DINER_DGRAD_W8=1 timeout 1500 python -m pytest tests/test_train_gpu.py -x -q -m gpu -s 2>&1 | grep -E "training path|conditioned|Frobenius|beyond|shipped|passed|failed|Error|error|assert" | tee $O/pytest_train.log | cut -c1-250
for v in 1 0 1 0; do
  echo "== DINER_DGRAD_W8=$v" | tee -a $O/time.txt
  DINER_DGRAD_W8=$v timeout 600 python tools/time_train.py --objects 1 --rays 4096 --steps 6 2>&1 | grep -E "rays x" | cut -c1-130 | tee -a $O/time.txt
done
DINER_DGRAD_W8=1 timeout 600 python tools/time_train.py --objects 4 --rays 4096 --steps 4 2>&1 | grep -E "rays x" | cut -c1-130 | tee -a $O/time.txt
DINER_DGRAD_W8=0 timeout 600 python tools/time_train.py --objects 4 --rays 4096 --steps 4 2>&1 | grep -E "rays x" | cut -c1-130 | tee -a $O/time.txt
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
DINER_DGRAD_W8=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o t -- python $R/tools/time_train.py --objects 1 --rays 4096 --steps 4 > $R/$O/prof.log 2>&1
f=$(find $R/$O/prof -name "*kernel_stats.csv" | head -1)
grep -E "k_wgrad512_w8|k_run512_f16x3|k_dgrad512_w8" $f | cut -c1-160 | tee -a $R/$O/time.txt
find $R/$O -name "*.db" -delete; find $R/$O -name "*trace.csv" -delete
