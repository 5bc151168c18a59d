#!/bin/bash
# round 4, call AD: pixel splits of the 16-group Winograd weight-gradient launch (kbench conv3 --pass wgrad alone; the option forces every TN launch, so alone only)
set -u
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
for sp in 0 2 4 8 16; do echo "## CG_TN_SPLITS=$sp"; CG_TN_SPLITS=$sp timeout 120 python scripts/kbench.py 128 --only conv3,conv2 --pass wgrad 2>/dev/null | grep "conv"; done | tee gpurun_out/ad_tn_splits.txt
