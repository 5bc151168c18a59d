#!/bin/bash
# round 4, call O: the MFMA localisation kernels at the 64x64 discriminator's branch shape (S 16, 64 planes): parity, config #5 A/B
set -u
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_full.py -m gpu -x -q -p no:cacheprovider -k "spatial_transformer or c5 or at_64 or 64" > gpurun_out/o_pytest.log 2>&1
echo "== pytest: $(grep -h ' passed\| failed' gpurun_out/o_pytest.log | tail -1)"; grep -h "^E " gpurun_out/o_pytest.log | head -8
BENCH_ARGS="--config 5" STEPS=30 bash scripts/gpu_ab_env.sh "CG_LOCNET_V1=1" "CG_LOCNET_V1=0" 2>&1 | tee gpurun_out/o_sweep.txt
