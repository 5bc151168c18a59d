#!/bin/bash
# parity file(s) + same-box A/B of environment switches:  QUICK_TESTS="..." bash scripts/gpu_quick_ab.sh "VAR=a" "VAR=b" ...
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 1200 python -m pytest ${QUICK_TESTS:-tests/test_gpu_parity.py tests/test_abi_step.py} -m gpu -x -q -p no:cacheprovider > gpurun_out/q_pytest.log 2>&1
echo "== pytest: $(grep -h ' passed\| failed' gpurun_out/q_pytest.log | tail -1) $(grep -h '^FAILED' gpurun_out/q_pytest.log | tr '\n' ' ')"
grep -h "^E " gpurun_out/q_pytest.log | head -12
[ $# -gt 0 ] && bash scripts/gpu_ab_env.sh "$@"
