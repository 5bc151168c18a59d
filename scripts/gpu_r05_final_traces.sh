#!/bin/bash
# Round-5 closing traces on the final build (the trace half of gpu_evidence_r05.sh PART=b without the bench lines, PMC passes and the lab)
set -u
cd "${GRAFT_REPO_ROOT:-.}"; ROOTD=$PWD; TAG=r05; mkdir -p gpurun_out; export TMPDIR=/tmp
: > gpurun_out/${TAG}_roofline_launch_durations.txt
for k in conv2 conv1 dconv2 conv3; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$ROOTD/gpurun_out/prof_k$k" -o k$k -- python "$ROOTD/scripts/kbench.py" 128 --only $k > "$ROOTD/gpurun_out/${TAG}_kbench_$k.txt" 2>/dev/null)
  f=$(find gpurun_out/prof_k$k -name "*kernel_trace.csv" | head -1)
  echo "## python scripts/kbench.py 128 --only $k   ($(grep -h "ups\|@" gpurun_out/${TAG}_kbench_$k.txt | tail -1))" >> gpurun_out/${TAG}_roofline_launch_durations.txt
  [ -n "$f" ] && python scripts/trace_by_grid.py "$f" igemm wino wgrad_reduce nn_splitk >> gpurun_out/${TAG}_roofline_launch_durations.txt
  rm -rf gpurun_out/prof_k$k
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$ROOTD/gpurun_out/prof_kw22" -o kw22 -- python "$ROOTD/scripts/wino22_bench.py" 128 512 256 8 10 > "$ROOTD/gpurun_out/${TAG}_wino22_layer.txt" 2>/dev/null)
f=$(find gpurun_out/prof_kw22 -name "*kernel_trace.csv" | head -1)
echo "## python scripts/wino22_bench.py 128 512 256 8 10   (G.conv2 in F(2x2,2x2) beside the direct kernels; 3 warm-up + 10 timed launches each)" >> gpurun_out/${TAG}_roofline_launch_durations.txt
[ -n "$f" ] && python scripts/trace_by_grid.py "$f" igemm wino wgrad_reduce >> gpurun_out/${TAG}_roofline_launch_durations.txt
rm -rf gpurun_out/prof_kw22
TAG=${TAG} bash scripts/gpu_r05_trace.sh
for c in 3 5; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$ROOTD/gpurun_out/prof_c$c" -o c$c -- python "$ROOTD/bench.py" --config $c --no-cpu-baseline --no-kernel-roofline --steps 8 --warmup 4 > /dev/null 2>&1)
  f=$(find gpurun_out/prof_c$c -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && python scripts/step_breakdown.py "$f" > gpurun_out/${TAG}_breakdown_config$c.txt 2>&1; head -3 gpurun_out/${TAG}_breakdown_config$c.txt
  rm -rf gpurun_out/prof_c$c
done
TAG=${TAG}graph BENCH_ARGS=--graph bash scripts/gpu_r05_trace.sh
