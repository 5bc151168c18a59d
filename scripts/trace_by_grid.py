#!/usr/bin/env python
"""Kernel trace -> (kernel, grid) groups: launches, average / min duration.  The isolated launches bench.py times for `roofline_top`
show up as large groups of one grid size, so their rocprofv3 durations can be read next to bench.py's HIP-event timings.
Usage: trace_by_grid.py <kernel_trace.csv> [name substring ...]"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
want = sys.argv[2:]
agg = collections.defaultdict(list)
for r in rows:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    n = re.sub(r"\(.*", "", n).replace("void ", "")
    if want and not any(w in n for w in want):
        continue
    grid = "x".join(str(r.get(k, "")) for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z")) if "Grid_Size_X" in r else str(r.get("Grid_Size", ""))
    agg[(n, grid)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print(f"{'kernel':62s} {'grid (threads)':>18s} {'launches':>8s} {'avg us':>9s} {'min us':>9s}")
for (n, grid), d in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    if len(d) >= int(__import__("os").environ.get("MINL", "8")):
        print(f"{n[:62]:62s} {grid:>18s} {len(d):8d} {sum(d) / len(d):9.1f} {min(d):9.1f}")
