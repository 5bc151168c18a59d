#!/bin/bash
# round 4, call AC: wgrad_lag per net (2 = only the net with the spatial transformers: D; 3 = only the generator), configs #2 and #3
set -u
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
{ STEPS=30 bash scripts/gpu_ab_env.sh "CG_WGRAD_LAG=0" "CG_WGRAD_LAG=2" "CG_WGRAD_LAG=3"
  BENCH_ARGS="--config 3" STEPS=20 bash scripts/gpu_ab_env.sh "CG_WGRAD_LAG=0" "CG_WGRAD_LAG=2" "CG_WGRAD_LAG=3"; } 2>&1 | tee gpurun_out/ac_sweep.txt
