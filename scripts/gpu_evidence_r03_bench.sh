#!/bin/bash
# second half of the round-3 evidence: the default bench command under rocprofv3 (kernel stats + per-grid durations of the isolated
# roofline launches), the un-profiled bench line, smoke()
set -u
cd "${GRAFT_REPO_ROOT:-.}"; ROOTD=$PWD
TAG=${TAG:-r03}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "captured or reproducible or graph_replay" > gpurun_out/q_pytest.log 2>&1
echo "== pytest: $(grep -h ' passed\| failed' gpurun_out/q_pytest.log | tail -1)"; grep -h "^E " gpurun_out/q_pytest.log | head -5
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
echo "== default bench under rocprofv3 --kernel-trace --stats"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOTD/gpurun_out/prof_${TAG}d" -o ${TAG}d -- python "$ROOTD/bench.py" --no-cpu-baseline > "$ROOTD/gpurun_out/${TAG}_bench_line_traced.json" 2> "$ROOTD/gpurun_out/prof_bench_default.log"); echo "rc=$?"
g=$(find gpurun_out/prof_${TAG}d -name "*kernel_stats.csv" | head -1); [ -n "$g" ] && cp "$g" gpurun_out/${TAG}_bench_default_kernel_stats.csv
f=$(find gpurun_out/prof_${TAG}d -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python scripts/trace_by_grid.py "$f" igemm_nn igemm_tn wino_gemm > gpurun_out/${TAG}_roofline_launch_durations.txt && head -14 gpurun_out/${TAG}_roofline_launch_durations.txt
rm -rf gpurun_out/prof_${TAG}d
echo "== bench"; timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/bench.err; echo "rc=$?"; cut -c1-600 gpurun_out/${TAG}_bench_line.json; tail -3 gpurun_out/bench.err
