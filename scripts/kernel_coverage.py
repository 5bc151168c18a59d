#!/usr/bin/env python
"""Which kernels of a bench run also ran inside the oracle-comparing GPU tests?

usage: kernel_coverage.py <bench kernel_stats.csv | kernel_trace.csv> <pytest kernel_stats.csv | kernel_trace.csv>
Both files come from `rocprofv3 --kernel-trace --stats --output-format csv`; kernel names are compared with their
template arguments (igemm_nn_kernel<128, 64, 2, 2, true, true, 16> != <64, 128, ...>), argument lists stripped."""
import csv, re, sys


def names(path):
    out = {}
    for r in csv.DictReader(open(path)):
        n = r.get("Name") or r.get("Kernel_Name")
        n = re.sub(r"\(anonymous namespace\)::", "", n)
        n = re.sub(r"\(.*", "", n).replace("void ", "").strip()
        out[n] = out.get(n, 0) + int(r.get("Calls", 1) or 1)
    return out


bench, tests = names(sys.argv[1]), names(sys.argv[2])
own = lambda n: not n.startswith("__amd_rocclr") and "at::" not in n and "rccl" not in n.lower()
missing = [n for n in bench if own(n) and n not in tests]
print(f"{len([n for n in bench if own(n)])} engine kernels in the bench run, {len(missing)} of them absent from the test run")
for n in sorted(bench):
    if own(n):
        print(f"{'MISSING' if n not in tests else 'covered':8s} bench calls {bench[n]:6d}  test calls {tests.get(n, 0):6d}  {n}")
sys.exit(1 if missing else 0)
