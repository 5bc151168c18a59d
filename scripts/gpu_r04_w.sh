#!/bin/bash
# round 4, call W: F(2x2,2x2) below the 2048-tile threshold (conv2 at N/2, conv1 at N) layer-alone; data gradient in 2 vs 4 K slices
set -u
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
{ echo "## conv2 at N/2 (1024 tiles)"; timeout 120 python scripts/wino22_bench.py 64 2>&1 | grep -v amdgpu.ids
  echo "## conv1 at N (512 tiles)"; timeout 120 python scripts/wino22_bench.py 128 512 512 4 2>&1 | grep -v amdgpu.ids
  for ks in 1 2; do echo "## conv2 at N, CG_WINO22_KSPLIT=$ks"; CG_WINO22_KSPLIT=$ks timeout 120 python scripts/wino22_bench.py 2>&1 | grep "data gradient"; done; } | tee gpurun_out/w_layer.txt
STEPS=30 bash scripts/gpu_ab_env.sh "CG_WINO22_KSPLIT=1" "CG_WINO22_KSPLIT=2" 2>&1 | tee gpurun_out/w_sweep.txt
