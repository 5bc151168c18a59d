#!/bin/bash
# round 5, call e: weight-gradient tile / split sweep on the shipping kernels (lab: 128x64 at half the splits = same GEMM time, half the partial sums)
mkdir -p gpurun_out/r05e
for kv in "X=0" "CG_TN_TILE=128064" "CG_TN_TILE=64128" "CG_TN_TARGET=2" "CG_TN_TARGET=4" "CG_TN_TILE=128064 CG_TN_TARGET=2" "CG_TN_TILE=128064 CG_TN_TARGET=4"; do
  echo "== $kv"; env $kv python scripts/kbench.py 128 --pass wgrad 2>/dev/null | awk '{print $0}' | grep -v "^layer" | cut -c1-40,80-110
done > gpurun_out/r05e/kbench_wgrad.txt 2>&1
cat gpurun_out/r05e/kbench_wgrad.txt
STEPS=30 bash scripts/gpu_ab_env.sh "X=0" "CG_TN_TILE=128064" "CG_TN_TARGET=2" "CG_TN_TARGET=4" > gpurun_out/r05e/ab.txt 2>&1; cat gpurun_out/r05e/ab.txt
