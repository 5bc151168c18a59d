#!/bin/bash
# round 4, call P: F(2x2,2x2) forward for upsample -> 3x3 (G's 512 -> 256 layer): parity, the layer alone, step A/B
set -u
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_full.py -m gpu -x -q -p no:cacheprovider -k "winograd or generator or step or planned" > gpurun_out/p_pytest.log 2>&1
echo "== pytest: $(grep -h ' passed\| failed' gpurun_out/p_pytest.log | tail -1)"; grep -h "^E " gpurun_out/p_pytest.log | head -8
timeout 120 python scripts/wino22_bench.py 2>&1 | tee gpurun_out/p_layer.txt
timeout 120 python scripts/wino22_bench.py 256 2>&1 | tee -a gpurun_out/p_layer.txt
STEPS=30 bash scripts/gpu_ab_env.sh "CG_WINOGRAD22=0" "CG_WINOGRAD22=1" 2>&1 | tee gpurun_out/p_sweep.txt
BENCH_ARGS="--config 3" STEPS=30 bash scripts/gpu_ab_env.sh "CG_WINOGRAD22=0" "CG_WINOGRAD22=1" 2>&1 | tee -a gpurun_out/p_sweep.txt
