#!/bin/bash
# round 5, call m: 16-byte loads in the split reductions (nn_splitk_reduce_kernel<1, V4>, wgrad_reduce_small_body_v4, the read-once path of
# wgrad_reduce_kernel<UPS> for 3x3) against the build before them (_base = git archive of the previous commit, built in place): parity
# tests, the generator's weight gradients alone, the step
mkdir -p gpurun_out/r05m
timeout 900 python -m pytest tests/test_gpu_parity_full.py tests/test_gpu_parity.py -x -q -m gpu -k "not training_steps_gradients" 2>&1 | grep -E "passed|failed|^E  " | head -8
for d in _base . _base .; do (cd $d && python scripts/kbench.py 128 --only conv2,conv1 --pass wgrad 2>/dev/null | grep "G\." | sed "s|^|$d |"); done | tee gpurun_out/r05m/kbench.txt
for rep in 1 2 3; do
  for d in _base .; do
    (cd $d && python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-kernel-roofline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$d', round(j['ms_per_step'],4))")
  done
done | tee gpurun_out/r05m/step.txt
