#!/bin/bash
# round 4, call T: F(2x2,2x2)-domain weight gradient (36 strided TN GEMMs from the forward's V): parity, layer alone, step A/B (3 = fwd + dgrad, 7 = + wgrad)
set -u
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 1100 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_full.py tests/test_abi_step.py -m gpu -x -q -p no:cacheprovider -k "winograd or generator or step or G32" > gpurun_out/t_pytest.log 2>&1
echo "== pytest: $(grep -h ' passed\| failed' gpurun_out/t_pytest.log | tail -1)"; grep -h "^E " gpurun_out/t_pytest.log | head -8
timeout 120 python scripts/wino22_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/t_layer.txt
STEPS=30 bash scripts/gpu_ab_env.sh "CG_WINOGRAD22=3" "CG_WINOGRAD22=7" 2>&1 | tee gpurun_out/t_sweep.txt
BENCH_ARGS="--config 5" STEPS=20 bash scripts/gpu_ab_env.sh "CG_WINOGRAD22=3" "CG_WINOGRAD22=7" 2>&1 | tee -a gpurun_out/t_sweep.txt
