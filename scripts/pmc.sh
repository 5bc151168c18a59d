#!/bin/bash
# PMC counter passes (separate from tracing, per the gpurun rules) on one command.  usage: pmc.sh <tag> <cmd...>
set -u
cd "${GRAFT_REPO_ROOT:-.}"; ROOTD=$PWD
tag=$1; shift
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
           "GRBM_GUI_ACTIVE FETCH_SIZE" "GRBM_GUI_ACTIVE WRITE_SIZE"; do
  i=$((i+1))
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$ROOTD/gpurun_out/pmc/${tag}_p$i" -o p -- "$@" > "$ROOTD/gpurun_out/pmc/${tag}_p$i.log" 2>&1)
  echo "pass $i rc=$?"
done
python3 - "$ROOTD/gpurun_out/pmc" "$tag" <<'PY'
import csv, glob, sys, collections
d, tag = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in sorted(glob.glob(f"{d}/{tag}_p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")[:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] in ("SQ_WAVES", "SQ_ACTIVE_INST_ANY", "FETCH_SIZE", "WRITE_SIZE"):
            calls[(k, r["Counter_Name"])] += 1
for k, c in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0))[:8]:
    n = max(calls[(k, "SQ_WAVES")], 1)
    print(k, "launches", n)
    for name, v in sorted(c.items()):
        print(f"    {name:28s} {v / n:16.1f} per launch")
PY

# optional: JSON of one kernel for bench.py / profiles  (PMC_JSON_KERNEL=<substring> [PMC_JSON_GRID=<threads>])
if [ -n "${PMC_JSON_KERNEL:-}" ]; then
  python3 "$ROOTD/scripts/pmc_json.py" "$ROOTD/gpurun_out/pmc" "$tag" "$PMC_JSON_KERNEL" ${PMC_JSON_GRID:-} > "$ROOTD/gpurun_out/pmc/${tag}.json"
  cat "$ROOTD/gpurun_out/pmc/${tag}.json"
fi
