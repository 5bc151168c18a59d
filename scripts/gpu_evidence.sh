#!/bin/bash
# The round's evidence in one gpurun call (needs ~6 GPU-minutes): part A = PMC passes of the roofline kernels, eager and
# graph-replay kernel traces, bench lines; part B = kernel trace of the GPU test-suite (coverage of the bench's kernels).
# TAG names the outputs (copy them from gpurun_out/ to profiles/ afterwards).
set -u
cd "${GRAFT_REPO_ROOT:-.}"
bash scripts/gpu_evidence_a.sh
bash scripts/gpu_evidence_b.sh
