#!/bin/bash
# round 4, call C: the MFMA localisation kernels (locnet.hip v2): parity subset, the launches alone v1 vs v2, the step with either;
# stream-priority experiment for the weight-gradient streams
set -u
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "spatial_transformer or discriminator or plan_options or planned or reproducible or per_module" > gpurun_out/c_pytest.log 2>&1
echo "== pytest: $(tail -1 gpurun_out/c_pytest.log)"; grep -h "^E " gpurun_out/c_pytest.log | head -8
for rep in 1 2; do
  CG_LOCNET_V1=1 timeout 120 python scripts/locbench.py 128 50 2>&1 | grep -v "^$" | tail -3
  CG_LOCNET_V1=0 timeout 120 python scripts/locbench.py 128 50 2>&1 | grep -v "^$" | tail -3
done | tee gpurun_out/c_locbench.txt
for v in 1 0; do CG_LOCNET_V1=$v python scripts/dbench.py 128 50 2>/dev/null | tail -1; done | tee -a gpurun_out/c_locbench.txt
STEPS=60 bash scripts/gpu_ab_env.sh CG_LOCNET_V1=1 CG_LOCNET_V1=0 2>&1 | tee gpurun_out/c_ab.txt
STEPS=60 bash scripts/gpu_ab_env.sh "CG_WGRAD_PRIO=0" "CG_WGRAD_PRIO=2 CG_BENCH_STREAM_PRIO=-1" "CG_WGRAD_PRIO=0 CG_BENCH_STREAM_PRIO=-1" 2>&1 | tee gpurun_out/c_prio.txt
