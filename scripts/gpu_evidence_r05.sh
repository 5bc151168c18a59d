#!/bin/bash
# Round-5 evidence in one gpurun call (PART=a: GPU suite + smoke; PART=b: profiles): PMC passes of the roofline kernels on the final build;
# per-launch durations of the roofline launches (kernel trace of kbench, where every row IS one of those launches); the default bench under
# rocprofv3 (kernel stats, eager timeline, breakdown, small-kernel chains); traced breakdowns of configs #3 / #5; the hipGraph replay's
# timeline; bench lines of configs #2 / #3 / #5 WITH roofline + cpu_baseline; the GEMM lab.
set -u
cd "${GRAFT_REPO_ROOT:-.}"; ROOTD=$PWD
TAG=${TAG:-r05}
mkdir -p gpurun_out
export TMPDIR=/tmp
if [ "${PART:-a}" = "a" ]; then
  echo "== pytest -m gpu"; ( time timeout 1500 python -m pytest tests -m gpu -x -q -s -p no:cacheprovider --durations=40 ) > gpurun_out/${TAG}_pytest.log 2>&1; echo "rc=$?"
  grep -h " passed\| failed\|^real\|s call \|s setup " gpurun_out/${TAG}_pytest.log | tail -45; grep -h "^E " gpurun_out/${TAG}_pytest.log | head -5
  grep -h "^\[grad\]\|^\[adam\]\|^\[outliers\]" gpurun_out/${TAG}_pytest.log > gpurun_out/${TAG}_step_gradients_vs_oracle.txt 2>/dev/null
  python __graft_entry__.py smoke > gpurun_out/${TAG}_smoke.txt 2>&1; tail -3 gpurun_out/${TAG}_smoke.txt
  exit 0
fi
echo "== PMC"; bash scripts/pmc_kernels.sh $TAG 2>&1 | tail -9
cp gpurun_out/pmc/${TAG}_pmc_kernels.json gpurun_out/${TAG}_pmc_kernels.json 2>/dev/null
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
head -40 gpurun_out/${TAG}_roofline_launch_durations.txt
echo "== default bench under rocprofv3 --kernel-trace --stats"
TAG=${TAG} bash scripts/gpu_r05_trace.sh
for c in 3 5; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$ROOTD/gpurun_out/prof_c$c" -o c$c -- python "$ROOTD/bench.py" --config $c --no-cpu-baseline --no-kernel-roofline --steps 8 --warmup 4 > /dev/null 2>&1)
  f=$(find gpurun_out/prof_c$c -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && python scripts/step_breakdown.py "$f" > gpurun_out/${TAG}_breakdown_config$c.txt 2>&1; head -3 gpurun_out/${TAG}_breakdown_config$c.txt
  rm -rf gpurun_out/prof_c$c
done
echo "== hipGraph replay traced"
TAG=${TAG}graph BENCH_ARGS=--graph bash scripts/gpu_r05_trace.sh
echo "== bench lines"; timeout 900 python bench.py > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/bench.err; echo "rc=$?"; cut -c1-300 gpurun_out/${TAG}_bench_line.json
for c in 3 5; do timeout 900 python bench.py --config $c > gpurun_out/${TAG}_bench_config$c.json 2>/dev/null; cut -c1-200 gpurun_out/${TAG}_bench_config$c.json; done
echo "== GEMM lab"; ./tools/gemm_lab 8 > gpurun_out/${TAG}_gemm_lab.txt 2>&1; ./tools/gemm_lab 4 >> gpurun_out/${TAG}_gemm_lab.txt 2>&1; ./tools/gemm_lab 2 >> gpurun_out/${TAG}_gemm_lab.txt 2>&1; head -5 gpurun_out/${TAG}_gemm_lab.txt
