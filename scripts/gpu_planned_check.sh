#!/bin/bash
# GPU check of the planned executor: parity suite, then the bench in its launch modes (same box)
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 1700 python -m pytest tests -m gpu -q -p no:cacheprovider ${PYTEST_ARGS:-} > gpurun_out/planned_pytest.log 2>&1; echo "rc=$?"; tail -60 gpurun_out/planned_pytest.log
for mode in "" "--no-graph"; do
  timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-roofline $mode 2>> gpurun_out/planned_bench.err | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print(j['config']['launch'], j['ms_per_step'], j['value'])"
done
