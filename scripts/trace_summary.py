#!/usr/bin/env python
"""Per-step kernel breakdown from a rocprofv3 kernel trace of `bench.py --no-graph` (steady-state last 4 steps)."""
import collections, csv, re, sys
path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof/r01_kernel_trace.csv"
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "adam_k" in r["Kernel_Name"]]
a0, a1 = adam[-9], adam[-1]
seg = rows[a0 + 1:a1 + 1]
t0, t1 = int(rows[a0]["End_Timestamp"]), int(rows[a1]["End_Timestamp"])
wall = (t1 - t0) / 1e6
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg) / 1e6
print("4 steps: wall %.2f ms/step, kernel busy %.2f ms/step (%.1f%%), launches/step %d" % (wall / 4, busy / 4, 100 * busy / wall, len(seg) // 4))
agg = collections.defaultdict(lambda: [0, 0])
for r in seg:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); n = re.sub(r"\(.*", "", n).replace("void ", "")
    agg[n][0] += 1; agg[n][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
g = sum(t for n, (c, t) in agg.items() if "igemm" in n) / 4e6
print("GEMM kernels %.2f ms/step, everything else %.2f ms/step" % (g, busy / 4 - g))
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print("%-60s calls/step %5.1f  ms/step %7.3f" % (n[:60], c / 4, t / 4e6))
