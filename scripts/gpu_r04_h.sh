#!/bin/bash
# round 4, call H: reductions' workgroup cap A/B (same box), concurrent G forward on / off
set -u
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
for v in 0 8 32; do echo "CG_RED_WGS_PER_CU=$v kbench conv3: $(CG_RED_WGS_PER_CU=$v python scripts/kbench.py 128 --only conv3 2>/dev/null | grep conv3)"; done | tee gpurun_out/h_sweep.txt
STEPS=40 bash scripts/gpu_ab_env.sh "CG_RED_WGS_PER_CU=0" "CG_RED_WGS_PER_CU=8" "CG_RED_WGS_PER_CU=32" "CG_RED_WGS_PER_CU=0 CG_CONCURRENT_G=0" "CG_RED_WGS_PER_CU=8 CG_CONCURRENT_G=0" 2>&1 | tee -a gpurun_out/h_sweep.txt
for v in 1 0; do echo "CG_CONCURRENT_G=$v: $(CG_CONCURRENT_G=$v python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-roofline --config 3 2>/dev/null | python -c 'import json,sys; j=json.loads(sys.stdin.read()); print("config3", round(j["ms_per_step"],3))')"; done | tee -a gpurun_out/h_sweep.txt
