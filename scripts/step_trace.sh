#!/bin/bash
# Kernel trace of the eager bench -> per-step breakdown (step_breakdown.py), one step's timeline with the queue of every launch
# (graph_timeline.py) and the launch-bound chains (small_kernel_chains.py).   step_trace.sh TAG [bench args...]
set -u
cd "${GRAFT_REPO_ROOT:-.}"; ROOTD=$PWD
export TMPDIR=/tmp
tag=$1; shift
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d "$ROOTD/gpurun_out/prof_$tag" -o $tag -- bash -c "cd $ROOTD && exec python bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-kernel-roofline --no-other-configs --no-dp-dry-run --no-reference-order --no-sustained $*" > "$ROOTD/gpurun_out/prof_$tag.log" 2>&1 < /dev/null); echo "rocprofv3 rc=$?"
f=$(find "gpurun_out/prof_$tag" -name "*kernel_trace.csv" 2>/dev/null | head -1)
if [ -z "$f" ]; then echo "no kernel trace"; tail -5 "gpurun_out/prof_$tag.log"; exit 1; fi
python scripts/step_breakdown.py "$f" > "gpurun_out/${tag}_eager_breakdown.txt" 2>&1
python scripts/graph_timeline.py "$f" > "gpurun_out/${tag}_eager_timeline.txt" 2>&1
python scripts/small_kernel_chains.py "$f" > "gpurun_out/${tag}_small_kernel_chains.txt" 2>&1
head -70 "gpurun_out/${tag}_eager_breakdown.txt"
rm -rf "gpurun_out/prof_$tag"
