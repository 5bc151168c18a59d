#!/bin/bash
# round 4, call A: the weight-gradient streams (net.hip, option wgrad_stream): full GPU suite, same-box A/B against the in-line
# schedule and against normal-priority streams, replayed-step timeline of the new default
set -u
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/a_pytest.log 2>&1
echo "== pytest: $(tail -1 gpurun_out/a_pytest.log)"; grep -h "^E " gpurun_out/a_pytest.log | head -8
STEPS=60 bash scripts/gpu_ab_env.sh CG_WGRAD_STREAM=0 CG_WGRAD_STREAM=1 "CG_WGRAD_STREAM=1 CG_WGRAD_PRIO=0" 2>&1 | tee gpurun_out/a_ab.txt
BENCH_ARGS=--graph STEPS=60 bash scripts/gpu_ab_env.sh CG_WGRAD_STREAM=0 CG_WGRAD_STREAM=1 2>&1 | tee -a gpurun_out/a_ab.txt
TAG=a4 bash scripts/gpu_graphtrace.sh > gpurun_out/a_graph_replay_breakdown.txt 2>&1; head -12 gpurun_out/a_graph_replay_breakdown.txt
f=$(find gpurun_out/prof_a4 -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python scripts/graph_timeline.py "$f" > gpurun_out/a_graph_replay_timeline.txt 2>&1
rm -rf gpurun_out/prof_a4
