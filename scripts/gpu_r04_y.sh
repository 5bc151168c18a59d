#!/bin/bash
# round 4, call Y: F(2x2,3x3) data gradient in K slices (wino_dsplit), reductions flushed in front of the last localisation net (early_flush): parity subset + A/B
set -u
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "winograd" > gpurun_out/y_pytest.log 2>&1
echo "== pytest: $(grep -h ' passed\| failed' gpurun_out/y_pytest.log | tail -1)"; grep -h "^E " gpurun_out/y_pytest.log | head -8
STEPS=30 bash scripts/gpu_ab_env.sh "CG_WINO_DSPLIT=0 CG_EARLY_FLUSH=0" "CG_WINO_DSPLIT=1 CG_EARLY_FLUSH=0" "CG_WINO_DSPLIT=0 CG_EARLY_FLUSH=1" "CG_WINO_DSPLIT=1 CG_EARLY_FLUSH=1" "CG_WINO_DSPLIT=1 CG_EARLY_FLUSH=1 CG_WINO_DGRAD_KSLICES=4" 2>&1 | tee gpurun_out/y_sweep.txt
BENCH_ARGS="--config 3" STEPS=20 bash scripts/gpu_ab_env.sh "CG_WINO_DSPLIT=0" "CG_WINO_DSPLIT=1" 2>&1 | tee -a gpurun_out/y_sweep.txt
