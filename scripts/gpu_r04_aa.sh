#!/bin/bash
# round 4, call AA: cheap sweeps on the final build - hipGraph replay vs eager, element-wise / column-reduce grid caps, wgrad_lag again (configs #2, #3)
set -u
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
{ STEPS=30 bash scripts/gpu_ab_env.sh "X=0" "CG_EW_WGS_PER_CU=3" "CG_EW_WGS_PER_CU=2" "CG_COLREDUCE_WGS_PER_CU=2" "CG_WGRAD_LAG=1"
  BENCH_ARGS="--graph" STEPS=30 bash scripts/gpu_ab_env.sh "X=0"
  BENCH_ARGS="--config 3" STEPS=20 bash scripts/gpu_ab_env.sh "X=0" "CG_WGRAD_LAG=1" "CG_EW_WGS_PER_CU=2"; } 2>&1 | tee gpurun_out/aa_sweep.txt
