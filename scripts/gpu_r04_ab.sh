#!/bin/bash
# round 4, call AB: BN-backward statistics with fewer than one workgroup per CU (fewer slots to wait for beside the weight-gradient GEMMs)
set -u
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
{ STEPS=30 bash scripts/gpu_ab_env.sh "CG_BNBWD_WGS_DIV=1" "CG_BNBWD_WGS_DIV=2" "CG_BNBWD_WGS_DIV=4" "CG_BNBWD_WGS_DIV=8"
  BENCH_ARGS="--config 3" STEPS=20 bash scripts/gpu_ab_env.sh "CG_BNBWD_WGS_DIV=1" "CG_BNBWD_WGS_DIV=4"; } 2>&1 | tee gpurun_out/ab_sweep.txt
