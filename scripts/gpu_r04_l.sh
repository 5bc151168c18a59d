#!/bin/bash
# round 4, call L: weight gradients held back one layer (option wgrad_lag) + own zeroing kernel: parity subset, step A/B, D alone
set -u
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dp.py -m gpu -x -q -p no:cacheprovider -k "plan_options or reproducible or graph or discriminator or generator or result_neutral or checkpoint or replicas or strict" > gpurun_out/l_pytest.log 2>&1
echo "== pytest: $(grep -h ' passed\| failed' gpurun_out/l_pytest.log | tail -1)"; grep -h "^E " gpurun_out/l_pytest.log | head -8
STEPS=40 bash scripts/gpu_ab_env.sh "CG_WGRAD_LAG=0" "CG_WGRAD_LAG=1" 2>&1 | tee gpurun_out/l_sweep.txt
for v in 0 1; do echo "CG_WGRAD_LAG=$v: $(CG_WGRAD_LAG=$v python scripts/dbench.py 128 40 2>/dev/null | tail -1)"; done | tee -a gpurun_out/l_sweep.txt
BENCH_ARGS="--config 3" STEPS=30 bash scripts/gpu_ab_env.sh "CG_WGRAD_LAG=0" "CG_WGRAD_LAG=1" 2>&1 | tee -a gpurun_out/l_sweep.txt
