#!/bin/bash
# round 5, call c: both generator forwards side by side (OPT.concurrent_g_both): neutrality tests + same-box A/B
mkdir -p gpurun_out/r05c
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "side_by_side or beside_the_discriminator or graph_replay or smoke" > gpurun_out/r05c/tests.txt 2>&1
tail -5 gpurun_out/r05c/tests.txt
STEPS=30 bash scripts/gpu_ab_env.sh "CG_CONCURRENT_G_BOTH=0" "CG_CONCURRENT_G_BOTH=1" > gpurun_out/r05c/ab.txt 2>&1
cat gpurun_out/r05c/ab.txt
