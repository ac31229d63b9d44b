#!/bin/bash
# One parametrised GPU job (round 6: replaces the one-off tools/job_r0*.sh of rounds 4-5; the ones a profiles/ file cites by name are kept).
#   /usr/local/graft/bin/gpurun --timeout 2700 -- 'bash tools/job.sh <job> [args]'       (several jobs: 'bash tools/job.sh a; bash tools/job.sh b')
# Everything is written under gpurun_out/ (merged back by gpurun); copy what should be judged into profiles/.
#
#   tests [pytest args]        the -m gpu suite with --durations=0                      -> gpurun_out/tests.log
#   train-tests                tests/test_train_gpu.py                                   -> gpurun_out/train_tests.log
#   train-time [lib ...]       tools/time_train.py at SB 4 x 4096 (+ each libdiner_hip_<lib>.so, alternating twice)  -> gpurun_out/train_time.log
#   train-prof [rays] [obj]    rocprofv3 kernel trace of the training step (tools/prof_train.sh)
#   train-pmc [rays] [obj]     HBM bytes / MfmaUtil per kernel of the training step (tools/pmc_train.sh)
#   l512-prof [rows]           phase timer of lin512_body (needs libdiner_hip_l512prof.so: build_variant('l512prof', ['DINER_L512_PROF']))
#   bench [bench args]         the driver's command (default --gpus 1 --steps 20 --warmup 5) -> gpurun_out/bench_line.json + a digest on stdout
#   profile <tag> [bench args] rocprofv3 stats + PMC passes of the bench (tools/profile_round.sh)
#   smoke                      __graft_entry__.smoke()
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
job=$1; shift
case "$job" in
  tests)
    timeout 2400 python -m pytest tests -m gpu -q --durations=0 "$@" > gpurun_out/tests.log 2>&1; echo "rc=$?" >> gpurun_out/tests.log
    tail -60 gpurun_out/tests.log ;;
  train-tests)
    timeout 1500 python -m pytest tests/test_train_gpu.py -x -q --durations=0 "$@" > gpurun_out/train_tests.log 2>&1; echo "rc=$?" >> gpurun_out/train_tests.log
    tail -5 gpurun_out/train_tests.log ;;
  train-time)
    : > gpurun_out/train_time.log
    for rep in 1 2; do
      for lib in "" "$@"; do
        if [ -n "$lib" ]; then export DINER_AMD_LIB=$PWD/diner_amd/libdiner_hip_$lib.so; else unset DINER_AMD_LIB; fi
        echo "== lib ${lib:-default} (rep $rep)" >> gpurun_out/train_time.log
        timeout 600 python tools/time_train.py --objects 4 --rays 4096 >> gpurun_out/train_time.log 2>&1
      done
      [ $# -eq 0 ] && break
    done
    unset DINER_AMD_LIB
    grep -v amdgpu.ids gpurun_out/train_time.log ;;
  train-prof) bash tools/prof_train.sh "${1:-4096}" "${2:-4}" 2>&1 | tail -8; head -30 "gpurun_out/prof_train_${2:-4}x${1:-4096}/stats.md" ;;
  train-pmc) bash tools/pmc_train.sh "${1:-4096}" "${2:-4}" 2>&1 | tail -25 ;;
  l512-prof)
    DINER_AMD_LIB=$PWD/diner_amd/libdiner_hip_l512prof.so timeout 600 python tools/prof_l512.py "$@" > gpurun_out/prof_l512.log 2>&1
    grep -v amdgpu.ids gpurun_out/prof_l512.log ;;
  bench)
    [ $# -eq 0 ] && set -- --gpus 1 --steps 20 --warmup 5
    ( time python bench.py "$@" ) > gpurun_out/bench_line.json 2> gpurun_out/bench_stderr.log; echo "rc=$?" >> gpurun_out/bench_stderr.log
    tail -5 gpurun_out/bench_stderr.log
    python - <<'PY'
import json
l = json.loads(open("gpurun_out/bench_line.json").read().strip().splitlines()[-1])
print({k: l.get(k) for k in ("value", "ms_per_step", "fallback_launches", "energy")})
print("encode", l.get("encode"))
t = l.get("train") or {}
print("train", {k: t.get(k) for k in ("ms_per_step", "forward_ms", "backward_ms", "host_enqueue_ms", "tflops_fp32_equivalent", "frac", "peak_memory_gib", "batched")})
c = l.get("cpu_baseline") or {}
print("cpu", {k: c.get(k) for k in ("value", "cores", "host_cores", "host_threads")})
print("modes", {k: (v["rays_per_s"], v["fallback_launches"], v["roofline"]["frac"]) for k, v in (l.get("modes") or {}).items()})
print("configs", {k[:28] + k[-8:]: (v["rays_per_s"], v["fallback_launches"], v["roofline"]["frac"]) for k, v in (l.get("configs") or {}).items()})
print("roofline", l["roofline"]["frac"], l["roofline"]["avg_launch_ms"])
PY
    ;;
  profile) bash tools/profile_round.sh "$@" ;;
  smoke) python __graft_entry__.py smoke ;;
  *) echo "unknown job '$job' (see the header of tools/job.sh)"; exit 2 ;;
esac
