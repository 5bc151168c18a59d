#!/bin/bash
# Validation + A/B of one staging variant of the NN kernels (default: the LDS-direct loads, CG_NN_GLDS=1)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
P="${PFENV:-CG_NN_GLDS=1}"
K="${KSEL:-glds}"
echo "== forced variants [$K]"; timeout 600 python -m pytest tests/test_gpu_parity_full.py -q -x -p no:cacheprovider -k "forced_nn and $K" 2>&1 | tail -4
echo "== benchmarked-batch layers under [$P]"; env $P timeout 600 python -m pytest tests/test_gpu_parity_full.py -q -p no:cacheprovider -k "benchmarked_batch" 2>&1 | tail -4
for v in "CG_NN_PF=1" "$P"; do
  echo "=== kbench [$v]"; env $v timeout 300 python scripts/kbench.py 128 2>&1 | tail -19
done
echo "=== trace [$P]"; env $P CATGAN_LIB=$PWD/cat-generator_amd/lib/libcatgan_hip_exptrace.so timeout 200 python scripts/wg_trace.py 2>&1 | grep -E "^==|span|K loop|prologue|epilogue"
for v in "CG_NN_PF=1" "$P" "CG_NN_PF=1" "$P"; do
  echo "== step bench [$v]"
  env $v timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'], d.get('config', {}).get('launch'))"
done
