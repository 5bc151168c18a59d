#!/bin/bash
# round 4, call D: full GPU suite on the build with the MFMA localisation kernels; replayed-step and eager timelines; default bench line
set -u
cd "${GRAFT_REPO_ROOT:-.}"; ROOTD=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/d_pytest.log 2>&1
echo "== pytest: $(grep -h ' passed\| failed' gpurun_out/d_pytest.log | tail -1)"; grep -h "^E " gpurun_out/d_pytest.log | head -8
TAG=d4 BENCH_ARGS="--graph" bash scripts/gpu_graphtrace.sh > gpurun_out/d_graph_replay_breakdown.txt 2>&1; head -14 gpurun_out/d_graph_replay_breakdown.txt
f=$(find gpurun_out/prof_d4 -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python scripts/graph_timeline.py "$f" > gpurun_out/d_graph_replay_timeline.txt 2>&1
rm -rf gpurun_out/prof_d4
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$ROOTD/gpurun_out/prof_de" -o de -- python "$ROOTD/bench.py" --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-roofline > "$ROOTD/gpurun_out/prof_de.log" 2>&1)
f=$(find gpurun_out/prof_de -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python scripts/graph_timeline.py "$f" > gpurun_out/d_eager_timeline.txt 2>&1
rm -rf gpurun_out/prof_de
timeout 600 python bench.py > gpurun_out/d_bench_line.json 2> gpurun_out/d_bench_err.log; cut -c1-400 gpurun_out/d_bench_line.json
