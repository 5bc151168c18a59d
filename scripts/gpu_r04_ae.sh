#!/bin/bash
# round 4, call AE: gradBias partials of the Winograd layer from the dy transform (CG_WINO_BIAS_FUSE): parity subset + A/B on configs #2 / #3
set -u
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_full.py -m gpu -x -q -p no:cacheprovider -k "winograd or spatial_convolution or conv3 or step" > gpurun_out/ae_pytest.log 2>&1
echo "== pytest: $(grep -h ' passed\| failed' gpurun_out/ae_pytest.log | tail -1)"; grep -h "^E " gpurun_out/ae_pytest.log | head -8
{ STEPS=30 bash scripts/gpu_ab_env.sh "CG_WINO_BIAS_FUSE=0" "CG_WINO_BIAS_FUSE=1"
  BENCH_ARGS="--config 3" STEPS=20 bash scripts/gpu_ab_env.sh "CG_WINO_BIAS_FUSE=0" "CG_WINO_BIAS_FUSE=1"; } 2>&1 | tee gpurun_out/ae_sweep.txt
