#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== winograd variants + 5x5 layers under CG_WINO_GLDS=1"; CG_WINO_GLDS=1 timeout 45 python -m pytest tests/test_gpu_parity_full.py -q -p no:cacheprovider -k "forced_winograd or (benchmarked_batch and 5x5 and G32up)" 2>&1 | tail -4
for v in 0 1; do echo "=== kbench conv3 [CG_WINO_GLDS=$v]"; CG_WINO_GLDS=$v timeout 20 python scripts/kbench.py 128 --only conv3 2>&1 | tail -2 | head -1; done
