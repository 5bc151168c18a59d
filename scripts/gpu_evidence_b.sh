#!/bin/bash
# Evidence, part B: kernel trace of the GPU test-suite -> which of the bench's kernels occur inside oracle-comparing tests
set -u
cd "${GRAFT_REPO_ROOT:-.}"; ROOTD=$PWD
TAG=${TAG:-r02b}
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest trace"; (cd /tmp && timeout ${PYTEST_TIMEOUT:-1200} rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOTD/gpurun_out/prof_${TAG}_tests" -o tests -- python -m pytest "$ROOTD/tests" -m gpu -q -p no:cacheprovider --deselect "$ROOTD/tests/test_gpu_dp.py" ${PYTEST_K:+-k "$PYTEST_K"} > "$ROOTD/gpurun_out/pytest_traced.log" 2>&1); echo "rc=$?"; tail -3 gpurun_out/pytest_traced.log
h=$(find gpurun_out/prof_${TAG}_tests -name "*kernel_stats.csv" | head -1)
[ -n "$h" ] && cp "$h" gpurun_out/${TAG}_tests_kernel_stats.csv && python scripts/kernel_coverage.py profiles/${TAG}_bench_kernel_stats.csv "$h" > gpurun_out/${TAG}_parity_kernel_coverage.txt; head -3 gpurun_out/${TAG}_parity_kernel_coverage.txt; grep MISSING gpurun_out/${TAG}_parity_kernel_coverage.txt | head
rm -rf gpurun_out/prof_${TAG}_tests
echo "== DP tests"; timeout 120 python -m pytest tests/test_gpu_dp.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3
