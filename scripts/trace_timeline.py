#!/usr/bin/env python
"""One step of a `bench.py --no-graph` kernel trace as a timeline (start, gap to the previous kernel's end, duration)."""
import csv, re, sys
path = sys.argv[1]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "adam_k" in r["Kernel_Name"]]
a0, a1 = adam[-3], adam[-1]
seg = rows[a0 + 1:a1 + 1]
prev = t0 = int(rows[a0]["End_Timestamp"])
for r in seg:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); n = re.sub(r"\(.*", "", n).replace("void ", "")
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f  gap %6.1f  dur %7.1f  grid %8s wg %4s  %s" % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, r.get("Grid_Size_X", "?"), r.get("Workgroup_Size_X", "?"), n[:70]))
    prev = e
