#!/bin/bash
# Evidence run: PMC passes of the roofline kernels, kernel trace of the GPU test-suite (coverage of the bench's kernels),
# the bench line, and the eager kernel trace / per-step breakdown of the bench.  TAG names the outputs.
set -u
cd "${GRAFT_REPO_ROOT:-.}"; ROOTD=$PWD
TAG=${TAG:-r02}
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== PMC"; bash scripts/pmc_kernels.sh $TAG 2>&1 | tail -12
echo "== bench trace"; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOTD/gpurun_out/prof_$TAG" -o $TAG -- python "$ROOTD/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-graph --no-kernel-roofline > "$ROOTD/gpurun_out/prof_bench.log" 2>&1); echo "rc=$?"
f=$(find gpurun_out/prof_$TAG -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python scripts/trace_summary.py "$f" 80 > gpurun_out/${TAG}_per_step_breakdown.txt && head -12 gpurun_out/${TAG}_per_step_breakdown.txt
g=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$g" ] && cp "$g" gpurun_out/${TAG}_bench_kernel_stats.csv
echo "== pytest trace"; (cd /tmp && timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOTD/gpurun_out/prof_${TAG}_tests" -o tests -- python -m pytest "$ROOTD/tests" -m gpu -q -x -p no:cacheprovider --deselect "$ROOTD/tests/test_gpu_dp.py" > "$ROOTD/gpurun_out/pytest_traced.log" 2>&1); echo "rc=$?"; tail -3 gpurun_out/pytest_traced.log
h=$(find gpurun_out/prof_${TAG}_tests -name "*kernel_stats.csv" | head -1)
[ -n "$h" ] && cp "$h" gpurun_out/${TAG}_tests_kernel_stats.csv && python scripts/kernel_coverage.py gpurun_out/${TAG}_bench_kernel_stats.csv "$h" > gpurun_out/${TAG}_parity_kernel_coverage.txt; head -3 gpurun_out/${TAG}_parity_kernel_coverage.txt; grep MISSING gpurun_out/${TAG}_parity_kernel_coverage.txt | head
rm -f gpurun_out/prof_*/*.db; rm -rf gpurun_out/prof_${TAG}_tests/*trace.csv
echo "== bench"; cp gpurun_out/pmc/${TAG}_pmc_kernels.json profiles/${TAG}_pmc_kernels.json 2>/dev/null; timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/bench.err; echo "rc=$?"; cut -c1-400 gpurun_out/${TAG}_bench_line.json
