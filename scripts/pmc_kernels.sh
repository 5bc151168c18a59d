#!/bin/bash
# PMC passes (separate from tracing) of the kernels bench.py's `roofline_top` times -> gpurun_out/pmc/<tag>_pmc_kernels.json
set -u
cd "${GRAFT_REPO_ROOT:-.}"; ROOTD=$PWD
SRC=${1:-r03}
for k in conv2 dconv2 conv3; do
  bash scripts/pmc.sh k_$k python $ROOTD/scripts/kbench.py 128 --only $k > gpurun_out/pmc_k_$k.txt 2>&1
  tail -3 gpurun_out/pmc_k_$k.txt
done
# the forward launch of G's first convolution alone (bench.py's igemm_nng entry since round 4)
bash scripts/pmc.sh k_w22 python $ROOTD/scripts/wino22_bench.py 128 512 256 8 10 > gpurun_out/pmc_k_w22.txt 2>&1; tail -3 gpurun_out/pmc_k_w22.txt
bash scripts/pmc.sh k_conv1f python $ROOTD/scripts/kbench.py 128 --only conv1 --pass fwd > gpurun_out/pmc_k_conv1f.txt 2>&1; tail -3 gpurun_out/pmc_k_conv1f.txt
python3 - "$ROOTD" "$SRC" <<'PY'
import json, subprocess, sys
root, src = sys.argv[1], sys.argv[2]
want = {"nn64x128": ("k_conv1f", "igemm_nng_kernel<64, 128, 2, 2, 32"),
        "nn64x128_dgrad_conv2": ("k_conv2", "igemm_nng_kernel<64, 128, 2, 2, 32"),
        "tn128x128": ("k_conv2", "igemm_tng_kernel<128, 128, 2, 2>"),
        "nn128x64": ("k_dconv2", "igemm_nn_kernel<128, 64, 2, 2, true, true, 16"),
        "wino_g16": ("k_conv3", "wino_gemm_g_kernel<16, 16>"),
        "wino_g32": ("k_conv3", "wino_gemm_g_kernel<32, 16>"),
        "wino22_fwd": ("k_w22", "wino_gemm_g_kernel<32, 9>"),
        "wino22_dgrad": ("k_w22", "wino_gemm_g_kernel<16, 9>")}
out = {}
for key, (tag, pat) in want.items():
    j = json.loads(subprocess.check_output([sys.executable, f"{root}/scripts/pmc_json.py", f"{root}/gpurun_out/pmc", tag, pat]))
    j["source"] = f"profiles/{src}_pmc_kernels.json: scripts/pmc_kernels.sh = separate rocprofv3 --kernel-trace --pmc passes of `{'python scripts/wino22_bench.py 128 512 256 8 10' if tag == 'k_w22' else 'python scripts/kbench.py 128 --only ' + ('conv1 --pass fwd' if tag == 'k_conv1f' else tag[2:])}`"
    out[key] = j
json.dump(out, open(f"{root}/gpurun_out/pmc/{src}_pmc_kernels.json", "w"), indent=1)
for k, v in out.items():
    print(k, {n: v.get(n) for n in ("launches_sampled", "mfma_pipe_util", "valu_per_mfma", "hbm_bytes_per_launch_corrected")})
PY
