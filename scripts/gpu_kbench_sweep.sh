#!/bin/bash
# per-layer GEMM timings under the NN staging switches (CG_NN_GLDS 1 = default, 2, 3), then the replayed step's timeline
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
for v in 1 2 3; do
  echo "== CG_NN_GLDS=$v"; CG_NN_GLDS=$v timeout 300 python scripts/kbench.py 128 2>&1 | tee gpurun_out/kbench_glds$v.txt | tail -40
done
echo "== D.conv2 with the 64x64 tile"; CG_NN_TILE=64064 timeout 200 python scripts/kbench.py 128 --only dconv2 2>&1 | tail -5
echo "== graph replay trace"; TAG=swg bash scripts/gpu_graphtrace.sh > gpurun_out/sw_graph_replay_breakdown.txt 2>&1; head -60 gpurun_out/sw_graph_replay_breakdown.txt
f=$(find gpurun_out/prof_swg -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python scripts/graph_timeline.py "$f" > gpurun_out/sw_graph_replay_timeline.txt 2>&1
[ -n "$f" ] && python scripts/small_kernel_chains.py "$f" > gpurun_out/sw_small_kernel_chains.txt 2>&1
rm -rf gpurun_out/prof_swg
