#!/bin/bash
# A/B on one box: bench with the fusion off / on (and any extra variants passed as "NAME=VAL,..." strings)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk|fclk" | head -6
for v in "$@"; do
  echo "== $v"
  env $(echo $v | tr ',' ' ') timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'], d.get('config', {}).get('launch'))"
done
