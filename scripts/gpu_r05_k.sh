#!/bin/bash
# round 5, call k: lean strided (+ statistics) epilogue of igemm_nng_kernel: parity, G's forward layers alone, the step - against _base on the same box
mkdir -p gpurun_out/r05k
python -m pytest tests/test_gpu_parity_full.py tests/test_gpu_parity.py -x -q -m gpu -k "forced_nn or lean or full_conv or generator or statistics or epilogue" 2>&1 | tail -2
for d in _base . _base .; do echo "== $d"; (cd $d && python scripts/kbench.py 128 --quick --pass fwd 2>/dev/null | grep -v "^layer" | cut -c1-62); done | tee gpurun_out/r05k/kbench_fwd.txt
for rep in 1 2 3; do for d in _base .; do (cd $d && python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-roofline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$d', round(j['ms_per_step'],4))"); done; done | tee gpurun_out/r05k/ab3.txt
