#!/bin/bash
# round 5, call i (stage 2): G's weight-gradient stream (st4) and the host's side stream, D at (1,2,3,1,3)
mkdir -p gpurun_out/r05i
run() { env CG_QMAP="$1" CG_QMAP_T=$2 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-roofline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('QMAP=$1 T=$2', round(j['ms_per_step'],4))"; }
{
echo "== stage 2"
for g in 0 1 2 3; do for t in 0 1 2 3; do run "1,2,3,$g;1,2,3,1,3" $t; done; done
echo "== D's branch stream and pack stream variants at the best so far"
run "2,2,3,2;1,2,3,1,3" 3; run "3,2,3,2;1,2,3,1,3" 3; run "1,2,3,2;2,2,3,1,3" 3; run "1,2,3,2;3,2,3,1,3" 3; run "1,2,3,2;1,2,3,1,3" 3
} | tee gpurun_out/r05i/qmap_stage2.txt
