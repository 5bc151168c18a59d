#!/bin/bash
# round-2 first GPU call: new full-size parity tests, kernel coverage of the tests vs the bench, bench line
set -u
cd "${GRAFT_REPO_ROOT:-.}"; ROOTD=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
nproc; 
echo "== new parity tests"; timeout 1500 python -m pytest tests/test_gpu_parity_full.py -m gpu -q --tb=short -p no:cacheprovider -x -s > gpurun_out/pytest_full.log 2>&1; echo "pytest rc=$?"; grep -E "^\[grad\]|^\[adam\]|passed|failed|Error|error" gpurun_out/pytest_full.log | tail -60
echo "== bench"; timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench rc=$?"; tail -3 gpurun_out/bench.log | cut -c1-600; tail -5 gpurun_out/bench.err
