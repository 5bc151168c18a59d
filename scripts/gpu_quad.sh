#!/bin/bash
# One-box validation + A/B of the k-quad LDS layout (CG_NN_QUAD / CG_TN_QUAD / CG_WINO_QUAD): forced-variant parity tests,
# the benchmarked-batch layer tests under the quad defaults, per-layer kernel times and the step bench, old vs new.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
Q="CG_NN_QUAD=1 CG_TN_QUAD=2 CG_WINO_QUAD=1"
echo "== forced quad variants"; timeout 600 python -m pytest tests/test_gpu_parity_full.py -q -x -p no:cacheprovider -k "forced and quad" 2>&1 | tail -15
echo "== benchmarked-batch layers under the quad kernels"; env $Q timeout 600 python -m pytest tests/test_gpu_parity_full.py -q -p no:cacheprovider -k "benchmarked_batch" 2>&1 | tail -15
for v in "CG_NN_QUAD=0" "$Q" "CG_NN_QUAD=2 CG_TN_QUAD=2 CG_WINO_QUAD=1" "CG_NN_QUAD=1 CG_TN_QUAD=1 CG_WINO_QUAD=0"; do
  echo "=== kbench [$v]"; env $v timeout 300 python scripts/kbench.py 128 2>&1 | tail -19
done
for v in "CG_NN_QUAD=0" "$Q" "CG_NN_QUAD=1" "CG_TN_QUAD=2" "CG_WINO_QUAD=1" "CG_NN_QUAD=2 CG_TN_QUAD=2 CG_WINO_QUAD=1" "CG_NN_QUAD=0" "$Q"; do
  echo "== step bench [$v]"
  env $v timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'], d.get('config', {}).get('launch'))"
done
