#!/bin/bash
# one hipGraph-replayed step as a timeline (scripts/graph_timeline.py) - superseded by scripts/gpu_trace_only.sh, kept for the r02b files it produced
set -u
cd "${GRAFT_REPO_ROOT:-.}"; ROOTD=$PWD
export TMPDIR=/tmp
mkdir -p gpurun_out
for cfg in "base CG_NN_GLDS=0" "glds2 CG_NN_GLDS=2"; do
  set -- $cfg; TAG=$1; V=$2
  (cd /tmp && env $V timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$ROOTD/gpurun_out/prof_$TAG" -o $TAG -- python "$ROOTD/bench.py" --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-roofline > "$ROOTD/gpurun_out/prof_tl.log" 2>&1)
  f=$(find gpurun_out/prof_$TAG -name "*kernel_trace.csv" | head -1)
  python scripts/graph_timeline.py "$f" > gpurun_out/timeline_$TAG.txt
  rm -rf gpurun_out/prof_$TAG
  wc -l gpurun_out/timeline_$TAG.txt
done
