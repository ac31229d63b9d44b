#!/bin/bash
# A/B of library builds on the GPU box: bit-level hash of the field output, then the 256x256 timing of each build (tools/ablate.sh)
cd $GRAFT_REPO_ROOT
for n in "$@"; do
  if [ "$n" == "base" ]; then L=diner_amd/libdiner_hip.so; else L=diner_amd/libdiner_hip_$n.so; fi
  echo "== $n"; DINER_AMD_LIB=$PWD/$L python tools/field_hash.py 2>&1 | tail -3
done
for rep in 1 2; do tools/ablate.sh "$@"; done
