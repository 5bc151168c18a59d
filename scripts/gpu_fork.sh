#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in "CG_G_FORK=early" "CG_G_FORK=late" "CG_CONCURRENT_G=0" "CG_G_FORK=early" "CG_G_FORK=late"; do
  echo "== step bench [$v CG_TN_GLDS=1]"
  env $v CG_TN_GLDS=1 timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'], d.get('config', {}).get('launch'))"
done
