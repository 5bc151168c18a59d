#!/bin/bash
# the fused localisation launches (csrc/locnet.hip): transformer / planned-pass parity tests, then the step with CG_FUSE_LOCNET 0 / 1 / 2
cd "${GRAFT_REPO_ROOT:-.}"; ROOTD=$PWD; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -x -k "spatial_transformer or planned_pass or plan_options or reproducible or discriminator" 2>&1 | tail -4
for v in 1 0; do CG_FUSE_LOCNET=$v python scripts/dbench.py 128 30; done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOTD/gpurun_out/prof_d" -o d -- python "$ROOTD/scripts/dbench.py" 128 20 > /dev/null 2>&1)
g=$(find gpurun_out/prof_d -name "*kernel_stats.csv" | head -1); grep -i "locnet" "$g" | cut -c1-160
f=$(find gpurun_out/prof_d -name "*kernel_trace.csv" | head -1); python scripts/trace_by_grid.py "$f" locnet
rm -rf gpurun_out/prof_d
bash scripts/gpu_ab_env.sh CG_FUSE_LOCNET=0 CG_FUSE_LOCNET=1
