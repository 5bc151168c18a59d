#!/usr/bin/env python
"""Kernel trace of a long run -> which kernels take longer in the slow iterations?  Iterations are delimited by every second adam_k launch;
the slowest 3 % of the iterations are compared with the median ones, kernel by kernel (mean duration per launch, launches per iteration).
Usage: slow_window.py <kernel_trace.csv>"""
import collections, csv, re, sys
import numpy as np
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    n = re.sub(r"\(.*", "", n).replace("void ", "")
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n))
rows.sort()
adam = [i for i, r in enumerate(rows) if r[2].startswith("adam_k")]
ends = adam[1::2]                       # the G update closes an iteration
its = []
for a, b in zip(ends[:-1], ends[1:]):
    seg = rows[a + 1:b + 1]
    its.append((seg[-1][1] - rows[a][1], seg))
dur = np.array([d for d, _ in its]) / 1e6
med = np.median(dur)
order = np.argsort(dur)
slow = [i for i in range(len(its)) if dur[i] > 1.15 * med]
norm = [i for i in order[len(order) // 4: 3 * len(order) // 4]]
print(f"{len(its)} iterations, median {med:.3f} ms; {len(slow)} slower than 1.15 x median: indices {slow[:40]}, mean {dur[slow].mean() if slow else 0:.3f} ms")
def agg(idx):
    d = collections.defaultdict(list)
    for i in idx:
        for s, e, n in its[i][1]:
            d[n].append((e - s) / 1e3)
    return {k: (np.mean(v), len(v) / max(1, len(idx))) for k, v in d.items()}
A, B = agg(slow), agg(norm)
print(f"{'kernel':58s} {'slow us':>9s} {'normal us':>10s} {'ratio':>6s} {'n/iter':>7s}  extra us/iter")
out = []
for k in A:
    if k in B:
        out.append((A[k][0] * A[k][1] - B[k][0] * B[k][1], k, A[k][0], B[k][0], A[k][1]))
for ex, k, a, b, n in sorted(out, reverse=True)[:25]:
    print(f"{k[:58]:58s} {a:9.1f} {b:10.1f} {a / b:6.2f} {n:7.1f}  {ex:9.1f}")
def busy(idx):
    tot = []
    for i in idx:
        seg = sorted(its[i][1]); t = 0; cur_s, cur_e = seg[0][0], seg[0][1]
        for s, e, _ in seg[1:]:
            if s > cur_e: t += cur_e - cur_s; cur_s, cur_e = s, e
            else: cur_e = max(cur_e, e)
        t += cur_e - cur_s
        tot.append(t / 1e6)
    return np.mean(tot)
if slow:
    print(f"GPU busy (union of kernels) per iteration: slow {busy(slow):.3f} ms, normal {busy(norm):.3f} ms")
