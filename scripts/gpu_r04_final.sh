#!/bin/bash
# round 4, last call: the whole GPU suite, smoke and the default bench line on the final build
set -u
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q -s -p no:cacheprovider > gpurun_out/r04_pytest.log 2>&1; echo "pytest rc=$?"
grep -h " passed\| failed" gpurun_out/r04_pytest.log | tail -1; grep -h "^E " gpurun_out/r04_pytest.log | head -5
grep -h "^\[grad\]\|^\[adam\]\|^\[outliers\]" gpurun_out/r04_pytest.log > gpurun_out/r04_step_gradients_vs_oracle.txt 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04_smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r04_smoke.txt
timeout 900 python bench.py > gpurun_out/r04_bench_line.json 2> gpurun_out/bench.err; echo "bench rc=$?"; cut -c1-260 gpurun_out/r04_bench_line.json
