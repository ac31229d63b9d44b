#!/bin/bash
# final verification of the round: the whole GPU suite, smoke(), the driver's bench command, the default bench command
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04z; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/t_gpu_all.log 2>&1; echo "rc=$?" >> $O/t_gpu_all.log
tail -3 $O/t_gpu_all.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 2 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; echo "bench rc=$?"
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?"
timeout 900 python tools/time_train.py --objects 4 --rays 4096 --steps 3 2>&1 | grep "rays x" | tee $O/time_train.txt | cut -c1-170
cut -c1-200 $O/bench_driver_cmd.json
