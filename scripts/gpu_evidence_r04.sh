#!/bin/bash
# Round-4 evidence in one gpurun call: GPU parity suite; PMC passes of the roofline kernels on the final build; per-launch durations of
# the roofline launches (kernel trace of kbench, where every row IS one of those launches); the default bench under rocprofv3
# (kernel stats); eager step timeline + breakdown; bench lines of configs #2 / #3 / #5.
set -u
cd "${GRAFT_REPO_ROOT:-.}"; ROOTD=$PWD
TAG=${TAG:-r04}
mkdir -p gpurun_out
export TMPDIR=/tmp
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -x -q -s -p no:cacheprovider > gpurun_out/${TAG}_pytest.log 2>&1; echo "rc=$?"
  grep -h " passed\| failed" gpurun_out/${TAG}_pytest.log | tail -1; grep -h "^E " gpurun_out/${TAG}_pytest.log | head -5
  grep -h "^\[grad\]\|^\[adam\]\|^\[outliers\]" gpurun_out/${TAG}_pytest.log > gpurun_out/${TAG}_step_gradients_vs_oracle.txt 2>/dev/null
fi
if [ "${SKIP_PMC:-0}" != "1" ]; then
  echo "== PMC"; bash scripts/pmc_kernels.sh $TAG 2>&1 | tail -8
  cp gpurun_out/pmc/${TAG}_pmc_kernels.json gpurun_out/${TAG}_pmc_kernels.json 2>/dev/null
fi
echo "== roofline launches alone: kernel trace of kbench (12 launches per pass: 2 warm-up + 10 timed)"
: > gpurun_out/${TAG}_roofline_launch_durations.txt
for k in conv2 conv1 dconv2 conv3; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$ROOTD/gpurun_out/prof_k$k" -o k$k -- python "$ROOTD/scripts/kbench.py" 128 --only $k > "$ROOTD/gpurun_out/${TAG}_kbench_$k.txt" 2>/dev/null)
  f=$(find gpurun_out/prof_k$k -name "*kernel_trace.csv" | head -1)
  echo "## python scripts/kbench.py 128 --only $k   ($(grep -h "ups\|@" gpurun_out/${TAG}_kbench_$k.txt | tail -1))" >> gpurun_out/${TAG}_roofline_launch_durations.txt
  [ -n "$f" ] && python scripts/trace_by_grid.py "$f" igemm wino wgrad_reduce >> gpurun_out/${TAG}_roofline_launch_durations.txt
  rm -rf gpurun_out/prof_k$k
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$ROOTD/gpurun_out/prof_kw22" -o kw22 -- python "$ROOTD/scripts/wino22_bench.py" 128 512 256 8 10 > "$ROOTD/gpurun_out/${TAG}_wino22_layer.txt" 2>/dev/null)
f=$(find gpurun_out/prof_kw22 -name "*kernel_trace.csv" | head -1)
echo "## python scripts/wino22_bench.py 128 512 256 8 10   (G.conv2 in F(2x2,2x2) beside the direct kernels; 3 warm-up + 10 timed launches each)" >> gpurun_out/${TAG}_roofline_launch_durations.txt
[ -n "$f" ] && python scripts/trace_by_grid.py "$f" igemm wino wgrad_reduce >> gpurun_out/${TAG}_roofline_launch_durations.txt
rm -rf gpurun_out/prof_kw22
head -50 gpurun_out/${TAG}_roofline_launch_durations.txt
echo "== default bench under rocprofv3 --kernel-trace --stats"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOTD/gpurun_out/prof_${TAG}d" -o ${TAG}d -- python "$ROOTD/bench.py" --no-cpu-baseline --no-kernel-roofline --steps 20 --warmup 5 > "$ROOTD/gpurun_out/${TAG}_bench_line_traced.json" 2> "$ROOTD/gpurun_out/prof_bench_default.log"); echo "rc=$?"
g=$(find gpurun_out/prof_${TAG}d -name "*kernel_stats.csv" | head -1); [ -n "$g" ] && cp "$g" gpurun_out/${TAG}_bench_kernel_stats.csv
f=$(find gpurun_out/prof_${TAG}d -name "*kernel_trace.csv" | head -1)
if [ -n "$f" ]; then
  python scripts/graph_timeline.py "$f" > gpurun_out/${TAG}_eager_timeline.txt 2>&1
  python scripts/step_breakdown.py "$f" > gpurun_out/${TAG}_eager_breakdown.txt 2>&1; head -16 gpurun_out/${TAG}_eager_breakdown.txt
  python scripts/small_kernel_chains.py "$f" > gpurun_out/${TAG}_small_kernel_chains.txt 2>&1
fi
rm -rf gpurun_out/prof_${TAG}d
echo "== bench"; timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/bench.err; echo "rc=$?"; cut -c1-300 gpurun_out/${TAG}_bench_line.json
for c in 3 5; do timeout 300 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-roofline > gpurun_out/${TAG}_bench_config$c.json 2>/dev/null; cut -c1-200 gpurun_out/${TAG}_bench_config$c.json; done
echo "== last switches, same box"; STEPS=40 bash scripts/gpu_ab_env.sh "X=0" "CG_WINOGRAD22=0" "CG_WINOGRAD22=1" "CG_WINOGRAD22=7" "CG_WGRAD_STREAM=0" "CG_CONCAT_OVERLAP=0" "CG_LOCNET_V1=1" 2>&1 | tee gpurun_out/${TAG}_final_switches.txt
