#!/bin/bash
# round 5, call d: neutrality tests again; hipGraph replay traced (queue assignment of the forked nodes) beside eager, same box
mkdir -p gpurun_out/r05d
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "side_by_side or graph_replay" > gpurun_out/r05d/tests.txt 2>&1; tail -3 gpurun_out/r05d/tests.txt
STEPS=30 bash scripts/gpu_ab_env.sh "X=0" "BENCH_ARGS=--graph" > gpurun_out/r05d/ab.txt 2>&1; cat gpurun_out/r05d/ab.txt
TAG=r05g BENCH_ARGS=--graph bash scripts/gpu_r05_trace.sh
