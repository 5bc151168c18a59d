#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; ROOTD=$PWD; export TMPDIR=/tmp; mkdir -p gpurun_out
for v in 1 0; do
  CG_FUSE_LOCNET=$v python scripts/dbench.py 128 30
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOTD/gpurun_out/prof_d" -o d -- python "$ROOTD/scripts/dbench.py" 128 20 > /dev/null 2>&1)
g=$(find gpurun_out/prof_d -name "*kernel_stats.csv" | head -1); head -30 "$g" | cut -d, -f1-6 | cut -c1-150
rm -rf gpurun_out/prof_d
