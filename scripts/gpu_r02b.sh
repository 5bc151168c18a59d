#!/bin/bash
# GPU check: all GPU tests, bench line, eager kernel trace + per-step breakdown.  TAG names the outputs.
set -u
cd "${GRAFT_REPO_ROOT:-.}"; ROOTD=$PWD
TAG=${TAG:-r02b}
mkdir -p gpurun_out
export TMPDIR=/tmp
if [ "${TESTS:-1}" = "1" ]; then
echo "== pytest -m gpu"; timeout 1800 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x ${PYTEST_ARGS:-} > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/pytest_gpu.log
fi
echo "== bench"; timeout 600 python bench.py --steps 20 --warmup 5 ${BENCH_ARGS:-} > gpurun_out/${TAG}_bench.log 2> gpurun_out/bench.err; echo "bench rc=$?"; tail -1 gpurun_out/${TAG}_bench.log | cut -c1-700; tail -3 gpurun_out/bench.err
if [ "${PROF:-1}" = "1" ]; then
  echo "== rocprofv3"; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOTD/gpurun_out/prof_$TAG" -o $TAG -- python "$ROOTD/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-graph --no-kernel-roofline > "$ROOTD/gpurun_out/prof_bench.log" 2>&1); echo "rocprof rc=$?"
  f=$(find gpurun_out/prof_$TAG -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && python scripts/trace_summary.py "$f" 70 > gpurun_out/${TAG}_per_step_breakdown.txt && head -45 gpurun_out/${TAG}_per_step_breakdown.txt
  g=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$g" ] && cp "$g" gpurun_out/${TAG}_bench_kernel_stats.csv
  rm -f gpurun_out/prof_$TAG/*.db
fi
