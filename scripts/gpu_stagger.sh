#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
for sg in 0 8 16 24 40; do
  echo "=== trace [CG_NN_STAGGER=$sg]"; CG_NN_STAGGER=$sg CATGAN_LIB=$PWD/cat-generator_amd/lib/libcatgan_hip_exptrace.so timeout 200 python scripts/wg_trace.py 2>&1 | grep -E "^==|span|K loop"
done
for sg in 0 16; do
  echo "=== kbench [CG_NN_STAGGER=$sg]"; CG_NN_STAGGER=$sg timeout 300 python scripts/kbench.py 128 --only conv2,dconv2,b4,conv1,conv3 2>&1 | tail -6
done
