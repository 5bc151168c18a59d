#!/bin/bash
# round 5, call g: working tree vs _base/: traced step breakdowns (which kernel / launch differs?)
mkdir -p gpurun_out/r05g
ROOTD=$PWD
export TMPDIR=/tmp
for d in _base .; do
  tag=$(echo $d | tr -d './_'); tag=${tag:-new}
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$ROOTD/gpurun_out/prof_$tag" -o $tag -- python "$ROOTD/$d/bench.py" --no-cpu-baseline --no-kernel-roofline --steps 12 --warmup 5 > /dev/null 2>&1)
  f=$(find gpurun_out/prof_$tag -name "*kernel_trace.csv" | head -1)
  python scripts/step_breakdown.py "$f" > gpurun_out/r05g/breakdown_$tag.txt 2>&1
  python scripts/graph_timeline.py "$f" > gpurun_out/r05g/timeline_$tag.txt 2>&1
  rm -rf gpurun_out/prof_$tag
  head -3 gpurun_out/r05g/breakdown_$tag.txt
done
