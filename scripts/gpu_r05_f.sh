#!/bin/bash
# round 5, call f: full GPU suite on the pruned build + bench
mkdir -p gpurun_out/r05f
( time timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider ) > gpurun_out/r05f/pytest.log 2>&1; tail -8 gpurun_out/r05f/pytest.log
python bench.py --no-cpu-baseline --no-kernel-roofline > gpurun_out/r05f/bench.json 2>/dev/null; cut -c1-220 gpurun_out/r05f/bench.json
python __graft_entry__.py smoke 2>&1 | tail -3
