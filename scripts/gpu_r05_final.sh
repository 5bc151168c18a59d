#!/bin/bash
# Round-5 closing call on the final build: GPU suite + smoke (PART=a of gpu_evidence_r05.sh), then the three bench lines with the PMC
# fields read from the committed profiles/r05_pmc_kernels.json.
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
PART=a bash scripts/gpu_evidence_r05.sh | tail -6
timeout 900 python bench.py > gpurun_out/r05_bench_line.json 2> gpurun_out/bench.err; echo "rc=$?"; cut -c1-260 gpurun_out/r05_bench_line.json
for c in 3 5; do timeout 900 python bench.py --config $c > gpurun_out/r05_bench_config$c.json 2>/dev/null; cut -c1-200 gpurun_out/r05_bench_config$c.json; done
