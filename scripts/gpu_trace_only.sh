#!/bin/bash
# the hipGraph-replayed step's breakdown, timeline and small-kernel chains (no tests)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
TAG=tg bash scripts/gpu_graphtrace.sh > gpurun_out/t_graph_replay_breakdown.txt 2>&1; head -${HEADN:-6} gpurun_out/t_graph_replay_breakdown.txt
f=$(find gpurun_out/prof_tg -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python scripts/graph_timeline.py "$f" > gpurun_out/t_graph_replay_timeline.txt 2>&1
[ -n "$f" ] && python scripts/small_kernel_chains.py "$f" > gpurun_out/t_small_kernel_chains.txt 2>&1
rm -rf gpurun_out/prof_tg
