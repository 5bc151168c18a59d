#!/bin/bash
# PMC passes of the discriminator's planned forward + backward (scripts/dbench.py): the round-3 kernels that are not GEMMs
set -u
cd "${GRAFT_REPO_ROOT:-.}"; ROOTD=$PWD
bash scripts/pmc.sh dnet python $ROOTD/scripts/dbench.py 128 5 > gpurun_out/pmc_dnet.txt 2>&1
python3 - "$ROOTD" <<'PY'
import json, subprocess, sys
root = sys.argv[1]
out = {}
for key, pat in (("locnet_fwd", "locnet_fwd_k"), ("locnet_bwd", "locnet_bwd_k"), ("head_fwd", "head_fwd_k"), ("head_bwd", "head_bwd_k"),
                 ("concat_drop", "concat4_drop_v4k"), ("bilinear_bwd", "bilinear_bwd_det_k<16"), ("act_pool_fwd", "act_pool2_fwd_k<true>")):
    j = json.loads(subprocess.check_output([sys.executable, f"{root}/scripts/pmc_json.py", f"{root}/gpurun_out/pmc", "dnet", pat]))
    out[key] = j
    print(key, {n: (round(v, 1) if isinstance(v, float) else v) for n, v in j.items() if n in ("launches_sampled", "hbm_bytes_per_launch_corrected", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAVES", "GRBM_GUI_ACTIVE")})
json.dump(out, open(f"{root}/gpurun_out/pmc/r03_pmc_dnet_kernels.json", "w"), indent=1)
PY
