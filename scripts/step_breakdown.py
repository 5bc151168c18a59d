#!/usr/bin/env python
"""Kernel trace of the (eager or replayed) bench -> per-step breakdown of the last three steps: wall, GPU busy (union over the queues),
idle, launches, kernel time by name, and the six phases of one step (boundaries = the step's marker kernels).
Usage: step_breakdown.py <kernel_trace.csv>"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: re.sub(r"\(.*", "", re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])).replace("void ", "")
adam = [i for i, r in enumerate(rows) if "adam_k" in r["Kernel_Name"]]
a0, a1 = adam[-7], adam[-1]          # last 3 steps (2 adam_k per step)
seg = rows[a0 + 1:a1 + 1]
t0, t1 = int(rows[a0]["End_Timestamp"]), int(rows[a1]["End_Timestamp"])
wall = (t1 - t0) / 3e6
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in seg)
busy, cs, ce = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > ce:
        busy += ce - cs; cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
tot = sum(e - s for s, e in iv)
print(f"3 steps: wall {wall:.3f} ms/step, GPU busy (union) {busy / 3e6:.3f} ms/step, idle gaps {wall - busy / 3e6:.3f} ms/step, "
      f"sum of kernel durations {tot / 3e6:.3f} ms/step, launches/step {len(seg) // 3}")
agg = collections.defaultdict(lambda: [0, 0])
for r in seg:
    agg[name(r)][0] += 1; agg[name(r)][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
g = sum(t for n, (c, t) in agg.items() if "igemm" in n or "wino_gemm" in n) / 3e6
print(f"GEMM kernels {g:.3f} ms/step, everything else {tot / 3e6 - g:.3f} ms/step")
# phases of the LAST step: [adam(G) of the previous step .. sigmoid_fwd #1] generator forward on N/2; .. head_bwd #1: D forward;
# .. adam(D): D backward + Adam; .. (sigmoid_fwd #2): generator forward on N; .. sigmoid_bwd: D forward + data gradient; .. adam(G)
last = rows[adam[-3] + 1:adam[-1] + 1]
tt0 = int(rows[adam[-3]]["End_Timestamp"])
def first(pat, k=0, after=0):
    hits = [r for r in last if pat in r["Kernel_Name"] and int(r["Start_Timestamp"]) >= after]
    return hits[k] if len(hits) > k else None
E = lambda r: int(r["End_Timestamp"]); S = lambda r: int(r["Start_Timestamp"])
sf = [r for r in last if "sigmoid_fwd_k" in r["Kernel_Name"]]
hb = first("head_bwd_k"); ad = [r for r in last if "adam_k" in r["Kernel_Name"]]; sb = first("sigmoid_bwd_k")
marks = [("generator forward on N/2 (fake images)", tt0, E(sf[0])), ("D forward", E(sf[0]), S(hb)), ("D backward + Adam", S(hb), E(ad[0]))]
if len(sf) > 1 and E(sf[1]) > E(ad[0]):        # generator forward of the G step in line (default since round 4)
    marks += [("generator forward on N", E(ad[0]), E(sf[1])), ("D forward + data gradient (G step)", E(sf[1]), S(sb))]
else:                                           # ... or beside the D update (CG_CONCURRENT_G=1)
    marks += [("D forward + data gradient (G step; the generator forward ran beside the D update)", E(ad[0]), S(sb))]
marks += [("generator backward + Adam", S(sb), E(ad[1]))]
print("phases of the last step (ms): " + "; ".join(f"{n} {(b - a) / 1e6:.2f}" for n, a, b in marks) + f"; total {(E(ad[1]) - tt0) / 1e6:.2f}")
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    print("%-64s calls/step %5.1f  ms/step %7.3f  avg us %6.1f" % (n[:64], c / 3, t / 3e6, t / c / 1e3))
