#!/bin/bash
# round 4, call F: strided 16-group Winograd weight gradient (parity subset + A/B), split / grid heuristics sweep on D alone and on the step
set -u
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_full.py -m gpu -x -q -p no:cacheprovider -k "wino or ups or generator or forced or variants" > gpurun_out/f_pytest.log 2>&1
echo "== pytest: $(grep -h ' passed\| failed' gpurun_out/f_pytest.log | tail -1)"; grep -h "^E " gpurun_out/f_pytest.log | head -8
for g in 4 16; do echo "CG_WINO_WGRAD_GROUPS=$g: $(CG_WINO_WGRAD_GROUPS=$g python scripts/kbench.py 128 --only conv3 2>/dev/null | grep conv3)"; done | tee gpurun_out/f_wino_wgrad.txt
echo "== D forward + backward alone (scripts/dbench.py 128 40)" | tee gpurun_out/f_sweep.txt
for kv in "X=0" "CG_SPLIT_TARGET=2" "CG_SPLIT_TARGET=2 CG_SPLIT_MINK=4" "CG_SPLIT_TARGET=4 CG_SPLIT_MINK=4" "CG_TN_TARGET=6" "CG_TN_TARGET=2" "CG_EW_WGS_PER_CU=4" "CG_EW_WGS_PER_CU=2" "CG_EW_WGS_PER_CU=16" "X=0"; do
  echo "$kv: $(env $kv python scripts/dbench.py 128 40 2>/dev/null | tail -1)"
done | tee -a gpurun_out/f_sweep.txt
echo "== step" | tee -a gpurun_out/f_sweep.txt
STEPS=40 bash scripts/gpu_ab_env.sh "X=0" "CG_WINO_WGRAD_GROUPS=4" "CG_SPLIT_TARGET=2 CG_SPLIT_MINK=4" "CG_EW_WGS_PER_CU=4" "CG_EW_WGS_PER_CU=2" "CG_TN_TARGET=6" 2>&1 | tee -a gpurun_out/f_sweep.txt
