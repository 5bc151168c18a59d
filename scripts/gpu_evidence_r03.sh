#!/bin/bash
# Round-3 evidence in one gpurun call: GPU parity suite, PMC passes of the roofline kernels (default build), the default bench
# under rocprofv3 (kernel stats + per-grid durations of the isolated roofline launches), graph-replay breakdown + timeline, bench lines.
set -u
cd "${GRAFT_REPO_ROOT:-.}"; ROOTD=$PWD
TAG=${TAG:-r03}
mkdir -p gpurun_out
export TMPDIR=/tmp
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -x -q -rP -p no:cacheprovider > gpurun_out/${TAG}_pytest.log 2>&1; echo "rc=$?"; tail -5 gpurun_out/${TAG}_pytest.log
  grep -h "^\[grad\]\|^\[adam\]" gpurun_out/${TAG}_pytest.log > gpurun_out/${TAG}_step_gradients_vs_oracle.txt 2>/dev/null
fi
if [ "${SKIP_PMC:-0}" != "1" ]; then
  echo "== PMC"; bash scripts/pmc_kernels.sh $TAG 2>&1 | tail -10
  cp gpurun_out/pmc/${TAG}_pmc_kernels.json profiles/${TAG}_pmc_kernels.json 2>/dev/null
fi
echo "== default bench under rocprofv3 --kernel-trace --stats"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOTD/gpurun_out/prof_${TAG}d" -o ${TAG}d -- python "$ROOTD/bench.py" --no-cpu-baseline > "$ROOTD/gpurun_out/${TAG}_bench_line_traced.json" 2> "$ROOTD/gpurun_out/prof_bench_default.log"); echo "rc=$?"
g=$(find gpurun_out/prof_${TAG}d -name "*kernel_stats.csv" | head -1); [ -n "$g" ] && cp "$g" gpurun_out/${TAG}_bench_default_kernel_stats.csv
f=$(find gpurun_out/prof_${TAG}d -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python scripts/trace_by_grid.py "$f" igemm_nn igemm_tn wino_gemm > gpurun_out/${TAG}_roofline_launch_durations.txt && head -12 gpurun_out/${TAG}_roofline_launch_durations.txt
rm -rf gpurun_out/prof_${TAG}d
echo "== graph replay trace"; TAG=${TAG}g bash scripts/gpu_graphtrace.sh > gpurun_out/${TAG}_graph_replay_breakdown.txt 2>&1; head -4 gpurun_out/${TAG}_graph_replay_breakdown.txt
f=$(find gpurun_out/prof_${TAG}g -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python scripts/graph_timeline.py "$f" > gpurun_out/${TAG}_graph_replay_timeline.txt 2>&1
[ -n "$f" ] && python scripts/small_kernel_chains.py "$f" > gpurun_out/${TAG}_small_kernel_chains.txt 2>&1
rm -rf gpurun_out/prof_${TAG}g
echo "== bench"; timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/bench.err; echo "rc=$?"; cut -c1-300 gpurun_out/${TAG}_bench_line.json
for c in 3 5; do timeout 300 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-roofline > gpurun_out/${TAG}_bench_config$c.json 2>/dev/null; cut -c1-200 gpurun_out/${TAG}_bench_config$c.json; done
