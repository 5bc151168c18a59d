#!/bin/bash
# repeat the replay-vs-eager / resume / repeated-pass tests in fresh processes per hand-over variant (scripts/build_handover.sh)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
for v in ${VARIANTS:-3 4 0}; do
  ok=0; bad=0
  for rep in $(seq 1 ${REPS:-8}); do
    CATGAN_LIB=$PWD/cat-generator_amd/lib/libcatgan_hip_ho$v.so timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "${K:-graph_replay_matches or resume_through or reproducible}" > gpurun_out/loop_${v}_${rep}.log 2>&1
    if grep -q " failed" gpurun_out/loop_${v}_${rep}.log; then bad=$((bad+1)); grep -h "^FAILED" gpurun_out/loop_${v}_${rep}.log | head -2; else ok=$((ok+1)); fi
  done
  echo "== handover $v: $ok runs passed, $bad failed"
done
