#!/bin/bash
# Practical fp32-MFMA ceiling (tools/mfma_peak), the LDS -> MFMA loop structures (tools/lds_mfma) and the LDS-direct-load
# semantics (tools/glds_test) on one box; the binaries come from __graft_entry__.build().
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
for t in ${TOOLS:-mfma_peak lds_mfma lds_mfma_agpr glds_test}; do echo "== $t"; timeout 120 ./tools/$t; done
