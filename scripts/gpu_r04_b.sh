#!/bin/bash
# round 4, call B: full GPU suite after the scratch-pool fix; EAGER kernel timelines of one step with the weight gradients in line
# (CG_WGRAD_STREAM=0) and on their own streams (=1), to see what the overlap does to the data-gradient chain
set -u
cd "${GRAFT_REPO_ROOT:-.}"; ROOTD=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/b_pytest.log 2>&1
echo "== pytest: $(tail -1 gpurun_out/b_pytest.log)"; grep -h "^E " gpurun_out/b_pytest.log | head -8
for v in 0 1; do
  (cd /tmp && CG_WGRAD_STREAM=$v timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$ROOTD/gpurun_out/prof_e$v" -o e$v -- python "$ROOTD/bench.py" --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-roofline > "$ROOTD/gpurun_out/prof_e$v.log" 2>&1)
  f=$(find gpurun_out/prof_e$v -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && python scripts/graph_timeline.py "$f" > gpurun_out/b_eager_timeline_wg$v.txt 2>&1
  rm -rf gpurun_out/prof_e$v
  tail -1 gpurun_out/b_eager_timeline_wg$v.txt
done
STEPS=60 bash scripts/gpu_ab_env.sh CG_WGRAD_STREAM=0 CG_WGRAD_STREAM=1 2>&1 | tee gpurun_out/b_ab.txt
