#!/bin/bash
# round 4, call Z: slice rule of the F(2x2,3x3) data gradient (two workgroups per CU, at most 4 slices) on configs #3, #5, #2
set -u
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
BENCH_ARGS="--config 3" STEPS=20 bash scripts/gpu_ab_env.sh "CG_WINO_DSPLIT=0" "CG_WINO_DGRAD_KSLICES=2" "CG_WINO_DSPLIT=1" 2>&1 | tee gpurun_out/z_sweep.txt
BENCH_ARGS="--config 5" STEPS=20 bash scripts/gpu_ab_env.sh "CG_WINO_DSPLIT=0" "CG_WINO_DSPLIT=1" 2>&1 | tee -a gpurun_out/z_sweep.txt
STEPS=30 bash scripts/gpu_ab_env.sh "CG_WINO_DSPLIT=0" "CG_WINO_DSPLIT=1" 2>&1 | tee -a gpurun_out/z_sweep.txt
