#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== small-batch parity under the defaults"; timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider 2>&1 | tail -8
echo "== DP test, old kernels"; CG_NN_GLDS=0 CG_TN_GLDS=0 timeout 300 python -m pytest tests/test_gpu_dp.py -m gpu -q -p no:cacheprovider -k "two_ranks" 2>&1 | grep -E "AssertionError:|passed|failed"
echo "== DP test, NN glds only"; CG_NN_GLDS=1 CG_TN_GLDS=0 timeout 300 python -m pytest tests/test_gpu_dp.py -m gpu -q -p no:cacheprovider -k "two_ranks" 2>&1 | grep -E "AssertionError:|passed|failed"
echo "== DP test, TN glds only"; CG_NN_GLDS=0 CG_TN_GLDS=1 timeout 300 python -m pytest tests/test_gpu_dp.py -m gpu -q -p no:cacheprovider -k "two_ranks" 2>&1 | grep -E "AssertionError:|passed|failed"
