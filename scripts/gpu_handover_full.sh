#!/bin/bash
# same-box A/B of the column reductions' hand-over variants (scripts/build_handover.sh): the operator / step parity file per variant
# (graph replay vs eager, resume and repeated-pass tests compare bits), then the step time
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
i=0
for v in ${VARIANTS:-3 4 3 4}; do
  i=$((i+1))
  export CATGAN_LIB=$PWD/cat-generator_amd/lib/libcatgan_hip_ho$v.so
  timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_abi_step.py -m gpu -q -p no:cacheprovider > gpurun_out/ho_full_${v}_$i.log 2>&1
  echo "== handover $v: $(grep -h ' passed\| failed' gpurun_out/ho_full_${v}_$i.log | tail -1) $(grep -h '^FAILED' gpurun_out/ho_full_${v}_$i.log | tr '\n' ' ')"
done
for rep in 1 2; do for v in ${BENCH_VARIANTS:-0 3 4}; do
  CATGAN_LIB=$PWD/cat-generator_amd/lib/libcatgan_hip_ho$v.so timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-kernel-roofline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('handover $v', round(j['ms_per_step'],4))"
done; done
