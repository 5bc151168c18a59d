#!/bin/bash
# round 4, call S: parity subset after the F(2x2,2x2) data gradient (default: split below one workgroup per CU)
set -u
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 1100 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_full.py tests/test_abi_step.py -m gpu -x -q -p no:cacheprovider -k "winograd or generator or step or G32" > gpurun_out/s_pytest.log 2>&1
echo "== pytest: $(grep -h ' passed\| failed' gpurun_out/s_pytest.log | tail -1)"; grep -h "^E " gpurun_out/s_pytest.log | head -8
