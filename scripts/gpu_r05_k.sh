#!/bin/bash
# round 5, call k: lean epilogue of igemm_tng_kernel: parity of every TN tile + the G layers' weight gradients against _base, same box
mkdir -p gpurun_out/r05k
python -m pytest tests/test_gpu_parity_full.py -x -q -m gpu -k "forced_tn or winograd_variants or full_conv" 2>&1 | tail -2
for d in _base . _base .; do echo "== $d"; (cd $d && python scripts/kbench.py 128 --quick --pass wgrad 2>/dev/null | grep -v "^layer" | cut -c1-40,80-110); done | tee gpurun_out/r05k/kbench_wgrad.txt
for rep in 1 2; do for d in _base .; do (cd $d && python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-roofline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$d', round(j['ms_per_step'],4))"); done; done | tee gpurun_out/r05k/ab.txt
