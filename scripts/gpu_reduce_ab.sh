#!/bin/bash
# wgrad_reduce_kernel after its LDS layout change: full-size + forced-variant weight-gradient tests, D forward + backward and the step, old vs new library
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity_full.py -m gpu -x -q -p no:cacheprovider -k "benchmarked_batch or forced_tn" > gpurun_out/q_pytest.log 2>&1
echo "== pytest: $(grep -h ' passed\| failed' gpurun_out/q_pytest.log | tail -1)"; grep -h "^E " gpurun_out/q_pytest.log | head -5
OLD=$PWD/cat-generator_amd/lib/libcatgan_hip_old.so
for rep in 1 2; do
  echo "old: $(CATGAN_LIB=$OLD python scripts/dbench.py 128 50 2>/dev/null | tail -1)"
  echo "new: $(python scripts/dbench.py 128 50 2>/dev/null | tail -1)"
done
for rep in 1 2; do for v in old new; do
  L=$PWD/cat-generator_amd/lib/libcatgan_hip.so; [ $v = old ] && L=$OLD
  CATGAN_LIB=$L timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-kernel-roofline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$v', round(j['ms_per_step'],4))"
done; done
