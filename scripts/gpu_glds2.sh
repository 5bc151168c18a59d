#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== forced variants [glds]"; timeout 600 python -m pytest tests/test_gpu_parity_full.py -q -x -p no:cacheprovider -k "forced_nn and glds" 2>&1 | tail -3
for v in "CG_NN_GLDS=0" "CG_NN_GLDS=1" "CG_NN_GLDS=2" "CG_NN_GLDS=0" "CG_NN_GLDS=1" "CG_NN_GLDS=2"; do
  echo "== step bench [$v]"
  env $v timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'], d.get('config', {}).get('launch'))"
done
CG_NN_GLDS=2 TAG=gglds2 bash scripts/gpu_graphtrace.sh > gpurun_out/gt_glds2.log 2>&1; rm -rf gpurun_out/prof_gglds2; head -16 gpurun_out/gt_glds2.log
