#!/bin/bash
# shader clock and package power while the step bench runs (is the step power / clock limited?)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in "CG_NN_GLDS=0" "CG_NN_GLDS=2"; do
  echo "== [$v]"
  ( env $v timeout 600 python bench.py --steps 1500 --warmup 5 --no-cpu-baseline --no-kernel-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('bench', d['ms_per_step'], d['value'])" ) &
  B=$!
  sleep 9
  for i in 1 2 3 4 5 6 7 8; do rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|Power \(W\)|junction|Sensor edge" | sed 's/GPU\[0\]\s*: //' | tr '\n' ';'; echo; sleep 0.7; done
  wait $B
done
