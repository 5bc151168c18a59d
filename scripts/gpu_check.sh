#!/bin/bash
# One gpurun call: smoke + GPU parity tests + short bench + rocprofv3 kernel trace.  Logs under gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== rocm-smi"; rocm-smi --showproductname 2>/dev/null | head -8
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -5 gpurun_out/smoke.log
echo "== pytest -m gpu"; timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -40 gpurun_out/pytest_gpu.log
echo "== bench"; timeout 600 python bench.py --steps ${STEPS:-10} --warmup 3 > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench rc=$?"; tail -3 gpurun_out/bench.log; tail -5 gpurun_out/bench.err
if [ "${PROF:-1}" = "1" ]; then
  echo "== rocprofv3"; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof" -o r01 -- python "$OLDPWD/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-graph --no-kernel-roofline > "$OLDPWD/gpurun_out/prof_bench.log" 2>&1); echo "rocprof rc=$?"
  find gpurun_out/prof -name "*stats*" | head; f=$(find gpurun_out/prof -name "*kernel_stats*.csv" | head -1); [ -n "$f" ] && head -40 "$f" | cut -c1-200
fi
