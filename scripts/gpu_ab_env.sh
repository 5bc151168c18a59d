#!/bin/bash
# same-box A/B of environment switches: gpu_ab_env.sh "VAR=a" "VAR=b" ...  (each setting run twice, interleaved)
# DRY=1: with the data-parallel dry run (bench.py --dp-dry-run 8) -> headline ms, dry-run ms, ratio
cd "${GRAFT_REPO_ROOT:-.}"
for rep in 1 2; do
for kv in "$@"; do
  if [ "${DRY:-0}" = 1 ]; then
  env $kv timeout 300 python bench.py --steps ${STEPS:-40} --warmup 10 --no-cpu-baseline --no-kernel-roofline --no-other-configs --no-reference-order --no-sustained 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); c=j['config']['collectives']; print('$kv', round(j['ms_per_step'],4), round(c['dry_run_ms_per_step'],4), round(c['dry_run_over_headline'],4))"
  continue; fi
  env $kv timeout 300 python bench.py --steps ${STEPS:-60} --warmup 10 --no-cpu-baseline --no-kernel-roofline --no-other-configs --no-dp-dry-run --no-reference-order --no-sustained ${BENCH_ARGS:-} 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$kv', j['config']['launch'], round(j['ms_per_step'],4), round(j['value'],1))"
done
done
