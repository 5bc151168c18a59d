#!/bin/bash
# Same-box validation + A/B of a kernel variant selected by environment options (CG_*), one gpurun call:
#   1. the forced-variant parity tests matching $KSEL (pytest -k),
#   2. the benchmarked-batch layer tests with the variant as the process default,
#   3. per-layer kernel times (scripts/kbench.py) for every setting in $SETTINGS,
#   4. the step bench for every setting, twice, alternating.
# Usage (from the repo root, as the gpurun command):
#   KSEL="forced_nn and glds" VARIANT="CG_NN_GLDS=1" SETTINGS="CG_NN_GLDS=0;CG_NN_GLDS=1" bash scripts/gpu_variant_ab.sh
# The r02b A/B logs in profiles/ came from this procedure:
#   quad layout   KSEL="forced and quad"       VARIANT="CG_NN_QUAD=1 CG_TN_QUAD=2 CG_WINO_QUAD=1"
#   prefetch 2    KSEL="forced_nn and pf2"     VARIANT="CG_NN_PF=2"            (+ scripts/wg_trace.py on the trace build)
#   LDS-direct NN KSEL="forced_nn and glds"    VARIANT="CG_NN_GLDS=1" (levels: SETTINGS="CG_NN_GLDS=0;CG_NN_GLDS=1;CG_NN_GLDS=2")
#   LDS-direct TN KSEL="forced_tn and glds"    VARIANT="CG_TN_GLDS=1"
#   Winograd      KSEL="forced_winograd"       VARIANT="CG_WINO_GLDS=1"        (KBENCH_ONLY=conv3)
#   fork point    STEPS_ONLY=1 SETTINGS="CG_G_FORK=early;CG_G_FORK=late;CG_CONCURRENT_G=0"
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
VARIANT="${VARIANT:-}"
SETTINGS="${SETTINGS:-CG_NN_GLDS=0;$VARIANT}"
if [ -z "${STEPS_ONLY:-}" ]; then
  if [ -n "${KSEL:-}" ]; then
    echo "== forced variants [$KSEL]"; timeout 600 python -m pytest tests/test_gpu_parity_full.py -q -x -p no:cacheprovider -k "$KSEL" 2>&1 | tail -4
  fi
  echo "== benchmarked-batch layers under [$VARIANT]"; env $VARIANT timeout 600 python -m pytest tests/test_gpu_parity_full.py -q -p no:cacheprovider -k "benchmarked_batch" 2>&1 | tail -4
  IFS=';' read -ra SS <<< "$SETTINGS"
  for v in "${SS[@]}"; do
    echo "=== kbench [$v]"; env $v timeout 300 python scripts/kbench.py 128 ${KBENCH_ONLY:+--only $KBENCH_ONLY} 2>&1 | tail -19
  done
fi
IFS=';' read -ra SS <<< "$SETTINGS"
for rep in 1 2; do
  for v in "${SS[@]}"; do
    echo "== step bench [$v]"
    env $v timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'], d.get('config', {}).get('launch'))"
  done
done
