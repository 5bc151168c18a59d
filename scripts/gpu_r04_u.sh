#!/bin/bash
# round 4, call U: eager timeline + breakdown of the step after the F(2x2,2x2) forward / data gradient
set -u
cd "${GRAFT_REPO_ROOT:-.}"; ROOTD=$PWD
mkdir -p gpurun_out; export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$ROOTD/gpurun_out/prof_u" -o u -- python "$ROOTD/bench.py" --no-cpu-baseline --no-kernel-roofline --steps 10 --warmup 5 > "$ROOTD/gpurun_out/u_bench.json" 2> "$ROOTD/gpurun_out/u_bench.log"); echo "rc=$?"
f=$(find gpurun_out/prof_u -name "*kernel_trace.csv" | head -1)
python scripts/graph_timeline.py "$f" > gpurun_out/u_eager_timeline.txt 2>&1
python scripts/step_breakdown.py "$f" > gpurun_out/u_eager_breakdown.txt 2>&1; head -12 gpurun_out/u_eager_breakdown.txt
rm -rf gpurun_out/prof_u
