#!/bin/bash
# round 5, call h: hardware-queue mapping of the streams (GPU_MAX_HW_QUEUES), same box; _base = the build before the pruning
mkdir -p gpurun_out/r05h
for rep in 1 2; do
  (cd _base && python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-roofline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('_base', round(j['ms_per_step'],4))")
  for q in 3 4 5; do
    if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
    python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-roofline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('new GPU_MAX_HW_QUEUES=$q', round(j['ms_per_step'],4))"
  done
  unset GPU_MAX_HW_QUEUES
done | tee gpurun_out/r05h/hwq.txt
