#!/bin/bash
# round 5: whole GPU suite + smoke on the current library (8-wave plain-fp16 kernel default, pass-dealt tile queues, generic path, ABI v5)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05r; mkdir -p $O
timeout 3400 python -m pytest tests -m gpu -q > $O/t_gpu_all.log 2>&1; echo "rc=$?" >> $O/t_gpu_all.log
tail -5 $O/t_gpu_all.log | cut -c1-250
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
