#!/bin/bash
# round 4, call Z3: the 5x5 Winograd layer alone (kbench conv3) at N and N/2 with its data gradient unsplit / in 2 / in 4 K slices; winograd parity
set -u
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "winograd or conv" > gpurun_out/z3_pytest.log 2>&1
echo "== pytest: $(grep -h ' passed\| failed' gpurun_out/z3_pytest.log | tail -1)"; grep -h "^E " gpurun_out/z3_pytest.log | head -8
for n in 128 64; do for ks in 1 2 4; do echo "## N=$n CG_WINO_DGRAD_KSLICES=$ks"; CG_WINO_DGRAD_KSLICES=$ks timeout 120 python scripts/kbench.py $n --only conv3 --pass dgrad 2>/dev/null | grep conv3; done; done | tee gpurun_out/z3_conv3_dgrad.txt
