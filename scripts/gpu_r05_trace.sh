#!/bin/bash
# eager step under rocprofv3 --kernel-trace: timeline, phase breakdown, small-kernel chains -> gpurun_out/${TAG}_*.txt   (env passes through)
set -u
cd "${GRAFT_REPO_ROOT:-.}"; ROOTD=$PWD
TAG=${TAG:-r05t}
mkdir -p gpurun_out
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOTD/gpurun_out/prof_${TAG}" -o ${TAG} -- python "$ROOTD/bench.py" --no-cpu-baseline --no-kernel-roofline --steps 12 --warmup 5 ${BENCH_ARGS:-} > "$ROOTD/gpurun_out/${TAG}_bench_line_traced.json" 2> "$ROOTD/gpurun_out/prof_${TAG}.log"); echo "rc=$?"
g=$(find gpurun_out/prof_${TAG} -name "*kernel_stats.csv" | head -1); [ -n "$g" ] && cp "$g" gpurun_out/${TAG}_bench_kernel_stats.csv
f=$(find gpurun_out/prof_${TAG} -name "*kernel_trace.csv" | head -1)
if [ -n "$f" ]; then
  python scripts/graph_timeline.py "$f" > gpurun_out/${TAG}_eager_timeline.txt 2>&1
  python scripts/step_breakdown.py "$f" > gpurun_out/${TAG}_eager_breakdown.txt 2>&1; head -4 gpurun_out/${TAG}_eager_breakdown.txt
  python scripts/small_kernel_chains.py "$f" > gpurun_out/${TAG}_small_kernel_chains.txt 2>&1
fi
rm -rf gpurun_out/prof_${TAG}
