#!/bin/bash
# round 5: GEMM-chain micro-benchmark (plain fp16): 64 / 96 / 128 columns per weight fragment, one or two waves per SIMD
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05d; mkdir -p $O
timeout 300 tools/ubench/bin/chain_f16 24 2>&1 | tee $O/chain_f16.txt | cut -c1-420
timeout 300 tools/ubench/bin/chain_f16 24 2>&1 | tee $O/chain_f16_again.txt | cut -c1-120
