#!/bin/bash
# kernel trace of the hipGraph-replayed bench (real inter-kernel gaps, unlike the eager trace).  The bench launches eagerly by
# default since round 3 (same kernels, same order); --graph keeps the tracer's per-launch cost out of the gaps.
set -u
cd "${GRAFT_REPO_ROOT:-.}"; ROOTD=$PWD
TAG=${TAG:-r02g}
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$ROOTD/gpurun_out/prof_$TAG" -o $TAG -- python "$ROOTD/bench.py" --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-roofline ${BENCH_ARGS:---graph} > "$ROOTD/gpurun_out/prof_graph.log" 2>&1); echo "rc=$?"
f=$(find gpurun_out/prof_$TAG -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "adam_k" in r["Kernel_Name"]]
a0, a1 = adam[-7], adam[-1]          # last 3 steps (2 adam per step)
seg = rows[a0 + 1:a1 + 1]
t0, t1 = int(rows[a0]["End_Timestamp"]), int(rows[a1]["End_Timestamp"])
wall = (t1 - t0) / 3e6
# union of busy intervals (streams overlap)
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in seg)
busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > cur_e: busy += cur_e - cur_s; cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = sum(e - s for s, e in iv)
print(f"graph replay, 3 steps: wall {wall:.3f} ms/step, GPU busy (union) {busy/3e6:.3f} ms/step, idle gaps {wall - busy/3e6:.3f} ms/step, sum of kernel durations {tot/3e6:.3f} ms/step, launches/step {len(seg)//3}")
agg = collections.defaultdict(lambda: [0, 0])
for r in seg:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); n = re.sub(r"\(.*", "", n).replace("void ", "")
    agg[n][0] += 1; agg[n][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
g = sum(t for n, (c, t) in agg.items() if "igemm" in n or "wino_gemm" in n) / 3e6
print(f"GEMM kernels {g:.3f} ms/step, everything else {tot/3e6 - g:.3f} ms/step")
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    print("%-60s calls/step %5.1f  ms/step %7.3f  avg us %6.1f" % (n[:60], c / 3, t / 3e6, t / c / 1e3))
PY
rm -f gpurun_out/prof_$TAG/*.db
