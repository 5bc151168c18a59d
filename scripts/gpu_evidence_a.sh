#!/bin/bash
# Evidence, part A: PMC passes of the roofline kernels, eager kernel trace / per-step breakdown, graph-replay breakdown, bench line.
set -u
cd "${GRAFT_REPO_ROOT:-.}"; ROOTD=$PWD
TAG=${TAG:-r02b}
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== PMC"; bash scripts/pmc_kernels.sh $TAG 2>&1 | tail -8
cp gpurun_out/pmc/${TAG}_pmc_kernels.json profiles/${TAG}_pmc_kernels.json 2>/dev/null
echo "== bench trace"; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOTD/gpurun_out/prof_$TAG" -o $TAG -- python "$ROOTD/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-graph --no-kernel-roofline > "$ROOTD/gpurun_out/prof_bench.log" 2>&1); echo "rc=$?"
f=$(find gpurun_out/prof_$TAG -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python scripts/trace_summary.py "$f" 80 > gpurun_out/${TAG}_per_step_breakdown.txt && head -4 gpurun_out/${TAG}_per_step_breakdown.txt
g=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$g" ] && cp "$g" gpurun_out/${TAG}_bench_kernel_stats.csv
rm -rf gpurun_out/prof_$TAG
echo "== graph replay trace"; TAG=${TAG}g bash scripts/gpu_graphtrace.sh > gpurun_out/${TAG}_graph_replay_breakdown.txt 2>&1; head -4 gpurun_out/${TAG}_graph_replay_breakdown.txt; rm -rf gpurun_out/prof_${TAG}g
echo "== default bench under rocprofv3 --stats"; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOTD/gpurun_out/prof_${TAG}d" -o ${TAG}d -- python "$ROOTD/bench.py" --no-cpu-baseline > "$ROOTD/gpurun_out/prof_bench_default.log" 2>&1); echo "rc=$?"
g=$(find gpurun_out/prof_${TAG}d -name "*kernel_stats.csv" | head -1); [ -n "$g" ] && cp "$g" gpurun_out/${TAG}_bench_default_kernel_stats.csv
rm -rf gpurun_out/prof_${TAG}d
echo "== bench"; timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/bench.err; echo "rc=$?"; cut -c1-300 gpurun_out/${TAG}_bench_line.json
for c in 3 5; do timeout 300 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-roofline > gpurun_out/${TAG}_bench_config$c.json 2>/dev/null; cut -c1-200 gpurun_out/${TAG}_bench_config$c.json; done
