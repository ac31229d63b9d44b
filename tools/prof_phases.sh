#!/bin/bash
# Phase timing of the per-view kernel (GPU box): needs diner_amd/libdiner_hip_prof.so = build_variant("prof", ["DINER_HN_PROF"]).
cd $GRAFT_REPO_ROOT
DINER_AMD_LIB=$PWD/diner_amd/libdiner_hip_prof.so python bench.py --steps 1 --warmup 1 --cpu-rays 0 --no-modes --no-configs --width 256 --height 256 2>&1 | grep "h3n prof" | tail -16
