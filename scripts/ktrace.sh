#!/bin/bash
# rocprofv3 kernel trace of one command -> per-(kernel, grid) launch counts and durations (scripts/trace_by_grid.py).
#   ktrace.sh TAG [name-substring,...] -- cmd...        output: gpurun_out/TAG_by_grid.txt (the raw trace is deleted)
set -u
cd "${GRAFT_REPO_ROOT:-.}"; ROOTD=$PWD
export TMPDIR=/tmp
tag=$1; shift
want=""
if [ "$1" != "--" ]; then want=$(echo "$1" | tr ',' ' '); shift; fi
shift
(cd /tmp && timeout "${KTRACE_TIMEOUT:-300}" rocprofv3 --kernel-trace --output-format csv -d "$ROOTD/gpurun_out/prof_$tag" -o $tag -- bash -c "cd $ROOTD && exec $*" > "$ROOTD/gpurun_out/prof_$tag.log" 2>&1 < /dev/null); echo "rocprofv3 rc=$?"
f=$(find "gpurun_out/prof_$tag" -name "*kernel_trace.csv" 2>/dev/null | head -1)
if [ -z "$f" ]; then echo "no kernel trace written"; tail -5 "gpurun_out/prof_$tag.log"; exit 1; fi
MINL=${MINL:-8} python scripts/trace_by_grid.py "$f" $want | tee "gpurun_out/${tag}_by_grid.txt"
rm -rf "gpurun_out/prof_$tag" "gpurun_out/prof_$tag.log"
