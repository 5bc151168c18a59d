#!/bin/bash
# round 5, call j: bn_act_bwd_stats_k with 16 loads in flight per lane: parity + same-box A/B against _base (the build before the pruning)
mkdir -p gpurun_out/r05j
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bn or batch_norm or generator or deterministic or reproduc" 2>&1 | tail -2
for rep in 1 2 3; do for d in _base .; do (cd $d && python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-roofline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$d', round(j['ms_per_step'],4))"); done; done | tee gpurun_out/r05j/ab_bnstats.txt
