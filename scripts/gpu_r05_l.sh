#!/bin/bash
# round 5, call l: batch-norm backward sums in the Winograd data gradient's epilogue (plan option bn_epilogue): parity + same-box A/B
mkdir -p gpurun_out/r05l
python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_full.py -x -q -m gpu -k "epilogue or generator or step or bit or deterministic or replay" 2>&1 | tail -3
STEPS=40 bash scripts/gpu_ab_env.sh "CG_NET_OPTIONS=bn_epilogue=0" "CG_NET_OPTIONS=bn_epilogue=1" | tee gpurun_out/r05l/ab.txt
BENCH_ARGS="--config 3" STEPS=30 bash scripts/gpu_ab_env.sh "CG_NET_OPTIONS=bn_epilogue=0" "CG_NET_OPTIONS=bn_epilogue=1" | tee gpurun_out/r05l/ab_c3.txt
