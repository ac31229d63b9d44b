#!/bin/bash
# round 5, first GPU call: (1) same-box A/B of the round-3 tree (ab_r03/, git archive f91ba99 built with its own sources) against HEAD on
# the headline workload, alternating, with rocm-smi power / sclk beside each run; (2) the new large-map oracle parity tests
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05a; mkdir -p $O
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-6s rays/s %8.0f  ms/frame %8.2f  pre %7.3f ms/launch  frac %.4f' % ('$1', d['value'], d['ms_per_step'], r.get('avg_launch_ms', -1), r['frac']))"; }
for i in 1 2 3; do
  ( cd ab_r03 && ../tools/power_sample.sh ../$O/pw_r03_$i.txt timeout 600 python bench.py --gpus 1 --steps 10 --warmup 2 --no-modes --no-configs --cpu-rays 0 2>/dev/null | grep '^{' | tail -1 > ../$O/line_r03_$i.json )
  line r03 < $O/line_r03_$i.json | tee -a $O/ab.txt; grep -E "power|sclk" $O/pw_r03_$i.txt | tr '\n' ' ' | tee -a $O/ab.txt; echo | tee -a $O/ab.txt
  tools/power_sample.sh $O/pw_head_$i.txt timeout 600 python bench.py --gpus 1 --steps 10 --warmup 2 --no-modes --no-configs --cpu-rays 0 2>/dev/null | grep '^{' | tail -1 > $O/line_head_$i.json
  line HEAD < $O/line_head_$i.json | tee -a $O/ab.txt; grep -E "power|sclk" $O/pw_head_$i.txt | tr '\n' ' ' | tee -a $O/ab.txt; echo | tee -a $O/ab.txt
done
timeout 1500 python -m pytest tests/test_large_maps_gpu.py -m gpu -x -q -s > $O/t_large.log 2>&1; echo "rc=$?" >> $O/t_large.log
grep -v "^$" $O/t_large.log | tail -40 | cut -c1-250
