#!/bin/bash
# round 4, call M: transpose tiles inside the batched (deferred) reduce: parity subset incl. a whole-step test, A/B
set -u
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_full.py tests/test_golden.py -m gpu -x -q -p no:cacheprovider -k "discriminator or linear or plan_options or reproducible or golden or (training_steps and c2-8) or adversarial or step" > gpurun_out/m_pytest.log 2>&1
echo "== pytest: $(grep -h ' passed\| failed' gpurun_out/m_pytest.log | tail -1)"; grep -h "^E " gpurun_out/m_pytest.log | head -8
STEPS=40 bash scripts/gpu_ab_env.sh "CG_WGRAD_TREDUCE=0" "CG_WGRAD_TREDUCE=1" 2>&1 | tee gpurun_out/m_sweep.txt
for v in 0 1; do echo "CG_WGRAD_TREDUCE=$v: $(CG_WGRAD_TREDUCE=$v python scripts/dbench.py 128 40 2>/dev/null | tail -1)"; done | tee -a gpurun_out/m_sweep.txt
