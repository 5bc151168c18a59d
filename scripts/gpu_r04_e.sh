#!/bin/bash
# round 4, call E: full GPU suite (outlier counts, strict transport, mixed-size loader), smoke, D forward+backward kernel trace
set -u
cd "${GRAFT_REPO_ROOT:-.}"; ROOTD=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider -s > gpurun_out/e_pytest.log 2>&1
echo "== pytest: $(grep -h ' passed\| failed' gpurun_out/e_pytest.log | tail -1)"; grep -h "^E " gpurun_out/e_pytest.log | head -8
grep -h "^\[outliers\]\|^\[grad\]" gpurun_out/e_pytest.log > gpurun_out/e_step_gradients_vs_oracle.txt; grep -c outliers gpurun_out/e_step_gradients_vs_oracle.txt
timeout 300 python __graft_entry__.py smoke 2>&1 | grep -i "smoke" | tee gpurun_out/e_smoke.txt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$ROOTD/gpurun_out/prof_dd" -o dd -- python "$ROOTD/scripts/dbench.py" 128 8 > "$ROOTD/gpurun_out/prof_dd.log" 2>&1)
f=$(find gpurun_out/prof_dd -name "*kernel_trace.csv" | head -1)
python - "$f" > gpurun_out/e_dbench_timeline.txt <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last forward + backward: from the last locnet_fwd2_k<16, 3> (first kernel of a forward after its weight packing) on
starts = [i for i, r in enumerate(rows) if "locnet_fwd" in r["Kernel_Name"] and "16, 3" in r["Kernel_Name"]]
i0 = starts[-1]
t0 = int(rows[i0]["Start_Timestamp"])
qs, ends = {}, []
for r in rows[i0:]:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); n = re.sub(r"\(.*", "", n).replace("void ", "")
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    q = qs.setdefault(r.get("Queue_Id", "?"), len(qs))
    conc = sum(1 for x in ends if x > s); ends.append(e)
    print("%8.1f -> %8.1f  dur %7.1f  q%d  +%d  grid %7s  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, conc, r.get("Grid_Size_X", "?"), n[:70]))
PY
rm -rf gpurun_out/prof_dd; tail -3 gpurun_out/e_dbench_timeline.txt
