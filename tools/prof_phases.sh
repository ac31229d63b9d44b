#!/bin/bash
# Phase timing of the per-view / post kernels (GPU box): needs diner_amd/libdiner_hip_prof.so = tools/ablate_h3n_build.sh prof:DINER_HN_PROF.
# usage: tools/prof_phases.sh [bench.py arguments; default: --width 256 --height 256]   (last launch of each kernel is printed)
cd $GRAFT_REPO_ROOT
ARGS="${@:---width 256 --height 256}"
DINER_AMD_LIB=$PWD/diner_amd/libdiner_hip_prof.so python bench.py --steps 1 --warmup 1 --cpu-rays 0 --no-modes --no-configs $ARGS 2>&1 | grep "h3n prof" | tail -34
