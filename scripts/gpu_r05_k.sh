#!/bin/bash
# round 5, call k: is the hardware-queue probe stable?  classes printed by five processes, step time of each, _base in between
mkdir -p gpurun_out/r05k
for rep in 1 2 3; do
  python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-roofline 2> gpurun_out/r05k/err.txt | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('.', round(j['ms_per_step'],4))"
  grep "cg: hardware" gpurun_out/r05k/err.txt
  (cd _base && python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-roofline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('_base', round(j['ms_per_step'],4))")
done | tee gpurun_out/r05k/probe_stability.txt
