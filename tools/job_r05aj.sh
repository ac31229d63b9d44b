#!/bin/bash
# round 5: training workspace in two parts (saved per object / shared work buffers) -- tests + timing + peak memory
O=gpurun_out/r05aj; mkdir -p $O
timeout 1800 python -m pytest tests/test_train_gpu.py tests/test_boundary_gpu.py -x -q -m gpu 2>&1 | tail -4 | tee $O/pytest_train.log | cut -c1-250
timeout 600 python tools/time_train.py --objects 4 --rays 4096 --steps 4 2>&1 | grep -E "rays x" | tee -a $O/time.txt | cut -c1-400
timeout 600 python tools/time_train.py --objects 1 --rays 4096 --steps 5 2>&1 | grep -E "rays x" | tee -a $O/time.txt | cut -c1-400
timeout 600 python tools/time_train.py --objects 4 --rays 128 --steps 20 2>&1 | grep -E "rays x" | tee -a $O/time.txt | cut -c1-400
