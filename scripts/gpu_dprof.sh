#!/bin/bash
# per-kernel time of D32_st3's planned forward + backward alone (scripts/dbench.py) under rocprofv3
cd "${GRAFT_REPO_ROOT:-.}"; ROOTD=$PWD; export TMPDIR=/tmp; mkdir -p gpurun_out
python scripts/dbench.py 128 30
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOTD/gpurun_out/prof_d" -o d -- python "$ROOTD/scripts/dbench.py" 128 20 > /dev/null 2>&1)
g=$(find gpurun_out/prof_d -name "*kernel_stats.csv" | head -1)
python - "$g" <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time per fwd+bwd: {tot / 23 / 1e3:.1f} us (23 passes traced)")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:45]:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Name"]); n = re.sub(r"\(.*", "", n).replace("void ", "")
    print("%-62s calls/pass %5.1f  us/pass %7.1f  avg us %7.1f" % (n[:62], int(r["Calls"]) / 23, float(r["TotalDurationNs"]) / 23 / 1e3, float(r["AverageNs"]) / 1e3))
PY
rm -rf gpurun_out/prof_d
