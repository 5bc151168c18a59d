#!/usr/bin/env python
"""Timeline of ONE hipGraph-replayed step from a rocprofv3 kernel trace: start / end relative to the step's first kernel, the
queue the kernel ran on, and how many other kernels were running when it started."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "adam_k" in r["Kernel_Name"]]
a0, a1 = adam[-3], adam[-1]          # the last whole step (2 adam_k per step)
seg = rows[a0 + 1:a1 + 1]
t0 = int(rows[a0]["End_Timestamp"])
qs = {}
ends = []
for r in seg:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); n = re.sub(r"\(.*", "", n).replace("void ", "")
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    q = qs.setdefault(r.get("Queue_Id", "?"), len(qs))
    conc = sum(1 for x in ends if x > s)
    ends.append(e)
    print("%8.1f -> %8.1f  dur %7.1f  q%d  +%d  grid %7s  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, conc, r.get("Grid_Size_X", r.get("Grid_Size", "?")), n[:64]))
