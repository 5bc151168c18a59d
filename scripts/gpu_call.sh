#!/bin/bash
# One parametrised GPU call (round 6: replaces the per-experiment gpu_rNN_x.sh one-shots).
#   gpurun --timeout T -- 'bash scripts/gpu_call.sh TAG "cmd 1" "cmd 2" ...'
# Every command runs from the repo root under its own `timeout` (CALL_TIMEOUT seconds, default 600); stdout + stderr of command i go to
# gpurun_out/TAG_i.log, whose last TAILN (default 25) lines are echoed, so that the call's verdict shows what happened.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
tag="$1"; shift
i=0
for cmd in "$@"; do
    i=$((i + 1))
    log="gpurun_out/${tag}_${i}.log"
    echo "== [$tag $i] $cmd"
    ( timeout "${CALL_TIMEOUT:-600}" bash -c "$cmd" ) > "$log" 2>&1 < /dev/null
    echo "   rc=$?"
    tail -n "${TAILN:-25}" "$log" | cut -c1-"${CUTW:-220}"
done
