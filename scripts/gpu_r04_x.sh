#!/bin/bash
# round 4, call X: the default bench line again (roofline entries timed on the capture stream) + smoke
set -u
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_line.json 2> gpurun_out/bench.err; echo "rc=$?"; cut -c1-260 gpurun_out/r04_bench_line.json
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04_smoke.txt 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/r04_smoke.txt
