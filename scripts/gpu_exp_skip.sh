#!/bin/bash
# upper bounds: step time with the localisation nets' launches dropped (CG_EXP_SKIP, timing only)
cd "${GRAFT_REPO_ROOT:-.}"
for v in 0 1 2 0 1 2; do
  CG_EXP_SKIP=$v timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-kernel-roofline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('CG_EXP_SKIP=$v', j['config']['launch'], round(j['ms_per_step'],4), round(j['value'],1))"
done
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -k "reproducible or planned_pass or plan_options or graph or resume or batchnorm" 2>&1 | tail -5
