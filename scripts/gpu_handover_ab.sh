#!/bin/bash
# same-box A/B of the column reductions' cross-workgroup hand-over (scripts/build_handover.sh): replay / resume / reproducibility tests
# three times each, then the step time
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
K="graph_replay or bit_reproducible or resume or batch_norm or bias"
for v in 1 2 0; do
  export CATGAN_LIB=$PWD/cat-generator_amd/lib/libcatgan_hip_ho$v.so
  for rep in 1 2 3; do
    echo "== handover $v rep $rep"; timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_abi_step.py -m gpu -q -p no:cacheprovider -k "$K" 2>&1 | tail -3
  done
done
for rep in 1 2; do for v in 1 2 0; do
  CATGAN_LIB=$PWD/cat-generator_amd/lib/libcatgan_hip_ho$v.so timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-kernel-roofline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('handover $v', round(j['ms_per_step'],4))"
done; done
