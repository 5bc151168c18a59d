#!/bin/bash
# Practical fp32-MFMA ceiling (tools/mfma_peak) and the LDS -> MFMA loop structures (tools/lds_mfma) on one box.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
for t in ${TOOLS:-mfma_peak lds_mfma lds_mfma_agpr}; do echo "== $t"; timeout 120 ./tools/$t; done
