#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== forced TN variants [glds]"; timeout 600 python -m pytest tests/test_gpu_parity_full.py -q -x -p no:cacheprovider -k "forced_tn and glds or forced_winograd" 2>&1 | tail -3
echo "== benchmarked-batch layers under [CG_TN_GLDS=1]"; CG_TN_GLDS=1 timeout 600 python -m pytest tests/test_gpu_parity_full.py -q -p no:cacheprovider -k "benchmarked_batch" 2>&1 | tail -3
for v in "CG_TN_GLDS=0" "CG_TN_GLDS=1"; do
  echo "=== kbench [$v]"; env $v timeout 300 python scripts/kbench.py 128 2>&1 | tail -19
done
for v in "CG_TN_GLDS=0" "CG_TN_GLDS=1" "CG_TN_GLDS=0" "CG_TN_GLDS=1"; do
  echo "== step bench [$v]"
  env $v timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'], d.get('config', {}).get('launch'))"
done
