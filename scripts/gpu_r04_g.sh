#!/bin/bash
# round 4, call G: parity subset after the grid-stride reductions; EW cap sweep; wgrad stream on/off at the new defaults
set -u
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_full.py -m gpu -x -q -p no:cacheprovider -k "not training_steps" > gpurun_out/g_pytest.log 2>&1
echo "== pytest: $(grep -h ' passed\| failed' gpurun_out/g_pytest.log | tail -1)"; grep -h "^E " gpurun_out/g_pytest.log | head -8
echo "kbench conv3: $(python scripts/kbench.py 128 --only conv3 2>/dev/null | grep conv3)" | tee gpurun_out/g_sweep.txt
STEPS=40 bash scripts/gpu_ab_env.sh "CG_EW_WGS_PER_CU=4" "CG_EW_WGS_PER_CU=3" "CG_EW_WGS_PER_CU=6" "CG_EW_WGS_PER_CU=8" "CG_WGRAD_STREAM=0" "CG_CONCURRENT_G=0" 2>&1 | tee -a gpurun_out/g_sweep.txt
