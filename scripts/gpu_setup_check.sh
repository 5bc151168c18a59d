#!/bin/bash
# GEMM set-up / epilogue changes: per-workgroup phase times (trace build), the variant + full-size parity tests, the step
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
CATGAN_LIB=$PWD/cat-generator_amd/lib/libcatgan_hip_exptrace.so timeout 300 python scripts/wg_trace.py --dconv2 2>/dev/null | grep -v "^   start\|^   end"
timeout 1200 python -m pytest tests/test_gpu_parity_full.py tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "${K:-lean_epilogue or forced_nn or benchmarked_batch or generic_gather or grouped or discriminator_forward or generator_forward or per_module or fused_chain}" > gpurun_out/q_pytest.log 2>&1
echo "== pytest: $(grep -h ' passed\| failed' gpurun_out/q_pytest.log | tail -1)"; grep -h "^E \|^FAILED" gpurun_out/q_pytest.log | head -8
for i in 1 2; do timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-kernel-roofline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('step', round(j['ms_per_step'],4))"; done
