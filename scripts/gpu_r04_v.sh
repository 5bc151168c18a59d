#!/bin/bash
# round 4, call V: the weight-gradient streams on a CU mask (3/4 or 1/2 of every XCD), the rest of the chip left to the data-gradient chain
set -u
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
STEPS=30 bash scripts/gpu_ab_env.sh "CG_WGRAD_CU_QUARTERS=0" "CG_WGRAD_CU_QUARTERS=3" "CG_WGRAD_CU_QUARTERS=2" 2>&1 | tee gpurun_out/v_sweep.txt
BENCH_ARGS="--config 3" STEPS=20 bash scripts/gpu_ab_env.sh "CG_WGRAD_CU_QUARTERS=0" "CG_WGRAD_CU_QUARTERS=3" 2>&1 | tee -a gpurun_out/v_sweep.txt
