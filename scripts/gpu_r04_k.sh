#!/bin/bash
# round 4, call K: transpose-reduce of the Winograd weight gradient + own zeroing kernel: parity subset, kbench, step A/B
set -u
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_full.py -m gpu -x -q -p no:cacheprovider -k "wino or ups or generator or result_neutral or reproducible or optim or adam" > gpurun_out/k_pytest.log 2>&1
echo "== pytest: $(grep -h ' passed\| failed' gpurun_out/k_pytest.log | tail -1)"; grep -h "^E " gpurun_out/k_pytest.log | head -8
for v in 0 1; do echo "CG_WGRAD_TREDUCE=$v kbench conv3: $(CG_WGRAD_TREDUCE=$v python scripts/kbench.py 128 --only conv3 2>/dev/null | grep conv3)"; done | tee gpurun_out/k_sweep.txt
STEPS=40 bash scripts/gpu_ab_env.sh "CG_WGRAD_TREDUCE=0" "CG_WGRAD_TREDUCE=1" 2>&1 | tee -a gpurun_out/k_sweep.txt
