#!/bin/bash
# same-box A/B of two library builds on the per-layer GEMM micro-benchmark and the step bench
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
BASE=cat-generator_amd/lib/libcatgan_hip_base.so
for rep in 1 2; do
echo "=== base (rep $rep)"; CATGAN_LIB=$PWD/$BASE python scripts/kbench.py 128 ${KARGS:-} 2>&1 | tail -22
echo "=== new (rep $rep)"; python scripts/kbench.py 128 ${KARGS:-} 2>&1 | tail -22
done
for v in base new base new; do
  if [ $v = base ]; then export CATGAN_LIB=$PWD/$BASE; else unset CATGAN_LIB; fi
  echo "== step bench $v"; timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])"
done
