#!/bin/bash
# round 4, call R: F(2x2,2x2) data gradient with its K slices (phases) split over blockIdx.z + a fixed-order sum: parity, layer alone, step A/B
set -u
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_full.py -m gpu -x -q -p no:cacheprovider -k "winograd or generator or step" > gpurun_out/r_pytest.log 2>&1
echo "== pytest: $(grep -h ' passed\| failed' gpurun_out/r_pytest.log | tail -1)"; grep -h "^E " gpurun_out/r_pytest.log | head -8
for ks in 0 1; do echo "CG_WINO22_KSPLIT=$ks"; CG_WINO22_KSPLIT=$ks timeout 120 python scripts/wino22_bench.py 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r_layer.txt
STEPS=30 bash scripts/gpu_ab_env.sh "CG_WINOGRAD22=1" "CG_WINOGRAD22=3 CG_WINO22_KSPLIT=0" "CG_WINOGRAD22=3 CG_WINO22_KSPLIT=1" 2>&1 | tee gpurun_out/r_sweep.txt
BENCH_ARGS="--config 5" STEPS=20 bash scripts/gpu_ab_env.sh "CG_WINOGRAD22=3 CG_WINO22_KSPLIT=0" "CG_WINOGRAD22=3 CG_WINO22_KSPLIT=1" 2>&1 | tee -a gpurun_out/r_sweep.txt
