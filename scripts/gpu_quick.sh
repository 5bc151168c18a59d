#!/bin/bash
# quick GPU check after a kernel change: the operator / module parity file, two bench lines, the replayed step's breakdown
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 900 python -m pytest ${QUICK_TESTS:-tests/test_gpu_parity.py} -m gpu -x -q -p no:cacheprovider > gpurun_out/q_pytest.log 2>&1
for i in 1 2; do timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-kernel-roofline 2>/dev/null | cut -c1-160; done
TAG=qg bash scripts/gpu_graphtrace.sh > gpurun_out/q_graph_replay_breakdown.txt 2>&1; head -${HEADN:-30} gpurun_out/q_graph_replay_breakdown.txt
f=$(find gpurun_out/prof_qg -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python scripts/graph_timeline.py "$f" > gpurun_out/q_graph_replay_timeline.txt 2>&1
rm -rf gpurun_out/prof_qg
grep -h "bn_act_bwd_stats\|colreduce4" gpurun_out/q_graph_replay_timeline.txt
echo "== pytest"; tail -3 gpurun_out/q_pytest.log
