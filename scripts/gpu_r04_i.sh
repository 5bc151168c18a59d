#!/bin/bash
# round 4, call I: new defaults (generator forward in line, reductions one workgroup per block): parity subset, remaining overlap switches
set -u
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "result_neutral or graph or reproducible or plan_options" > gpurun_out/i_pytest.log 2>&1
echo "== pytest: $(grep -h ' passed\| failed' gpurun_out/i_pytest.log | tail -1)"; grep -h "^E " gpurun_out/i_pytest.log | head -8
STEPS=40 bash scripts/gpu_ab_env.sh "X=0" "CG_CONCAT_OVERLAP=0" "CG_WGRAD_STREAM=0" "CG_EW_WGS_PER_CU=3" "CG_EW_WGS_PER_CU=2" "CG_PACK_OVERLAP=0" 2>&1 | tee gpurun_out/i_sweep.txt
BENCH_ARGS=--graph STEPS=40 bash scripts/gpu_ab_env.sh "X=0" 2>&1 | tee -a gpurun_out/i_sweep.txt
