#!/bin/bash
# round 5 final measurement pass: profiles (rocprofv3 stats + PMC) of the four workloads, the driver's bench command, the default command,
# a 2-rank self-launched line (ranks share the GPU), training step timing
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05zz; mkdir -p $O
CFG5="--facescape --width 1024 --height 1024 --samples 192"
timeout 900 tools/profile_round.sh r05zz_800 > $O/prof_800.log 2>&1
timeout 900 tools/profile_round.sh r05zz_800_f16 --precision f16 > $O/prof_800_f16.log 2>&1
timeout 900 tools/profile_round.sh r05zz_cfg5_f16 $CFG5 --precision f16 > $O/prof_cfg5_f16.log 2>&1
timeout 900 tools/profile_round.sh r05zz_cfg5_f16x3 $CFG5 --precision f16x3 > $O/prof_cfg5_f16x3.log 2>&1
python - <<'PY'
import json, glob
ent = {}
for t in ("800", "800_f16", "cfg5_f16", "cfg5_f16x3"):
    e = json.load(open(f"gpurun_out/prof_r05zz_{t}/pmc_latest_entry.json"))
    for k, v in e.items():
        ent[f"{k} @ {v['workload']}"] = v
json.dump(ent, open("profiles/pmc_latest.json", "w"), indent=1)
json.dump(ent, open("gpurun_out/r05zz/pmc_latest.json", "w"), indent=1)
print(json.dumps(ent, indent=1))
PY
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; echo "bench rc=$?"
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?"
timeout 900 python bench.py --gpus 2 --steps 3 --warmup 1 --no-configs > $O/bench_2ranks_one_gpu.json 2> $O/bench_2ranks.err; echo "bench 2 ranks rc=$?"
timeout 900 python bench.py --gpus 1 --force-dist --check-frame --steps 3 --warmup 1 --no-modes --no-configs --cpu-rays 0 > $O/bench_rccl_one_rank.json 2> $O/bench_rccl_one_rank.err; echo "bench rccl one rank rc=$?"
timeout 900 python tools/time_train.py --objects 4 --rays 4096 --steps 3 2>&1 | grep "rays x" | tee $O/time_train.txt | cut -c1-200
timeout 900 python tools/time_train.py --objects 1 --rays 4096 --steps 5 2>&1 | grep "rays x" | tee -a $O/time_train.txt | cut -c1-200
cut -c1-300 $O/bench_driver_cmd.json
