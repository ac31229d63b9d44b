#!/bin/bash
# item 6 experiment: block 0's first tap requests in front of the lin_in GEMM (f16x3), with rocm-smi power / clock beside each run
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04j; mkdir -p $O
for n in base g0early g0early4 base; do
  if [ "$n" == "base" ]; then L=diner_amd/libdiner_hip.so; else L=diner_amd/libdiner_hip_$n.so; fi
  echo "hash $n: $(DINER_AMD_LIB=$PWD/$L python tools/field_hash.py 2>/dev/null | head -1)" >> $O/ab.txt
  DINER_AMD_LIB=$PWD/$L tools/power_sample.sh $O/power_$n.txt python bench.py --steps 6 --warmup 1 --cpu-rays 0 --no-modes --no-configs > $O/line_$n.json 2>/dev/null
  python -c "import json,sys; d=json.loads([l for l in open('$O/line_$n.json') if l.startswith('{')][-1]); r=d['roofline']; print('%-10s rays/s %8.0f  pre %7.3f ms/launch  post total %7.2f ms  frac %.4f  whole %.4f' % ('$n', d['value'], r['avg_launch_ms'], r['post_kernel_ms_total'], r['frac'], r['whole_path']['frac']))" >> $O/ab.txt
  tr '\n' ' ' < $O/power_$n.txt >> $O/ab.txt; echo >> $O/ab.txt
done
cat $O/ab.txt
