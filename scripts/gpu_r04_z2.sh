#!/bin/bash
# round 4, call Z2: parity subset on the final slice rule (config #3's generator runs a 4-slice Winograd data gradient) + the bench lines of #3 / #5
set -u
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_full.py -m gpu -x -q -p no:cacheprovider -k "winograd or 256 or G32up or c3 or generator" > gpurun_out/z2_pytest.log 2>&1
echo "== pytest: $(grep -h ' passed\| failed' gpurun_out/z2_pytest.log | tail -1)"; grep -h "^E " gpurun_out/z2_pytest.log | head -8
for c in 3 5; do timeout 300 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-roofline > gpurun_out/r04_bench_config$c.json 2>/dev/null; cut -c1-200 gpurun_out/r04_bench_config$c.json; done
