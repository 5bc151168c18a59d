#!/usr/bin/env python
"""Latency-bound part of one hipGraph-replayed step: chains of kernels shorter than 10 us that run while NO kernel of 10 us or
more is on the chip.  Input: the rocprofv3 kernel trace scripts/gpu_graphtrace.sh leaves (last whole step = between adam_k launches).
Usage: small_kernel_chains.py <kernel_trace.csv> [threshold_us=10]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
thr = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 10e3
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "adam_k" in r["Kernel_Name"]]
a0, a1 = adam[-3], adam[-1]
seg = rows[a0 + 1:a1 + 1]
t0 = int(rows[a0]["End_Timestamp"])
name = lambda r: re.sub(r"\(.*", "", re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])).replace("void ", "").split("<")[0]
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name(r)) for r in seg]
big = [(s, e) for s, e, _ in ks if e - s >= thr]
small = [(s, e, n) for s, e, n in ks if e - s < thr and not any(bs < e and be > s for bs, be in big)]
chains, cur = [], []
for s, e, n in small:
    # a chain continues while no big kernel starts between its members
    if cur and not any(cur[-1][1] <= bs < s for bs, _ in big) and s - cur[-1][1] < thr:
        cur.append((s, e, n))
    else:
        if cur:
            chains.append(cur)
        cur = [(s, e, n)]
if cur:
    chains.append(cur)
tot = sum(c[-1][1] - c[0][0] for c in chains)
wall = int(rows[a1]["End_Timestamp"]) - t0
print(f"Chains of kernels shorter than {thr / 1e3:.0f} us that run while NO kernel of {thr / 1e3:.0f} us or more is on the chip (one hipGraph-replayed step under")
print(f"rocprofv3; a traced tiny kernel lasts ~5 us, un-traced ~2-3 us): the latency-bound part of the step's dependent chain.")
print(f"{sum(len(c) for c in chains)} of the step's {len(ks)} launches, {tot / 1e3:.0f} us of {wall / 1e3:.0f} us.\n")
for c in chains:
    if len(c) >= 2:
        print("%8.1f .. %8.1f us %3d launches %7.1f us  %s" % ((c[0][0] - t0) / 1e3, (c[-1][1] - t0) / 1e3, len(c), (c[-1][1] - c[0][0]) / 1e3,
                                                         ", ".join(n for _, _, n in c)))
