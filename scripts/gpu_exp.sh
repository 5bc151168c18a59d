#!/bin/bash
# Ablation timing of igemm_nn_kernel (libraries from scripts/build_exp.sh): which part of the K loop costs what.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
for e in ${EXPS:-0 1 2 4 6 7 8}; do
  if [ "$e" = 0 ]; then unset CATGAN_LIB; else export CATGAN_LIB=$PWD/cat-generator_amd/lib/libcatgan_hip_exp$e.so; fi
  echo "=== CG_EXP=$e"; timeout 200 python scripts/kbench.py 128 --only ${ONLY:-conv2,dconv2,b4,conv1} 2>&1 | tail -5 | head -4
done
