#!/bin/bash
# PMC passes (separate from tracing) of the kernels that carry the step -> gpurun_out/pmc/<tag>_pmc_kernels.json (round 6 set: the roofline
# launch of every configuration, the fused Winograd and skinny kernels, D's weight-gradient launches with their partial-sum volume)
set -u
cd "${GRAFT_REPO_ROOT:-.}"; ROOTD=$PWD
SRC=${1:-r06}
run() { tag=$1; shift; bash scripts/pmc.sh $tag "$@" > gpurun_out/pmc_$tag.txt 2>&1; tail -2 gpurun_out/pmc_$tag.txt; }
run k_conv2 python $ROOTD/scripts/kbench.py 128 --only conv2
run k_conv1f python $ROOTD/scripts/kbench.py 128 --only conv1 --pass fwd
run k_conv3 python $ROOTD/scripts/kbench.py 128 --only conv3
run k_dconv2 python $ROOTD/scripts/kbench.py 128 --only dconv2
run k_dconv2w python $ROOTD/scripts/kbench.py 128 --only dconv2,dbr16,b45,b4 --pass wgrad
run k_b4w python $ROOTD/scripts/kbench.py 128 --only b4 --pass wgrad
run k_b45w python $ROOTD/scripts/kbench.py 128 --only b45 --pass wgrad
run k_head python $ROOTD/scripts/kbench.py 128 --only dhead
run k_skinny python $ROOTD/scripts/kbench.py 128 --only gconv4,dconv1
run k_c3 python $ROOTD/scripts/kbench.py 256 --only conv3 --pass fwd
run k_c5 python $ROOTD/scripts/c5_wgrad.py
python3 - "$ROOTD" "$SRC" <<'PY'
import json, subprocess, sys
root, src = sys.argv[1], sys.argv[2]
want = {"tn128x128": ("k_conv2", "igemm_tng_kernel<128, 128, 2, 2>", "kbench.py 128 --only conv2"),
        "nn64x128": ("k_conv1f", "igemm_nng_kernel<64, 128, 2, 2, 32", "kbench.py 128 --only conv1 --pass fwd"),
        "wino_g16": ("k_conv3", "wino_gemm_g_kernel<16, 16>", "kbench.py 128 --only conv3"),
        "wino_g32": ("k_conv3", "wino_gemm_g_kernel<32, 16>", "kbench.py 128 --only conv3"),
        "wino3": ("k_dconv2", "wino3_fused_k", "kbench.py 128 --only dconv2"),
        "tn128x64_d_conv2_wgrad": ("k_dconv2", "igemm_tng_kernel<128, 64, 2, 2>", "kbench.py 128 --only dconv2"),
        "tn_d_wgrads": ("k_dconv2w", "igemm_tng_kernel<128, 64, 2, 2>", "kbench.py 128 --only dconv2,dbr16,b45,b4 --pass wgrad"),
        "tn128x128_d_wgrads": ("k_dconv2w", "igemm_tng_kernel<128, 128, 2, 2>", "kbench.py 128 --only dconv2,dbr16,b45,b4 --pass wgrad"),
        "tn128x128_d_7x7_wgrad_position_major": ("k_b4w", "igemm_tng_kernel<128, 128, 2, 2>", "kbench.py 128 --only b4 --pass wgrad"),
        "tn128x128_d_5x5_wgrad_position_major": ("k_b45w", "igemm_tng_kernel<128, 128, 2, 2>", "kbench.py 128 --only b45 --pass wgrad"),
        "head_wgrad": ("k_head", "head_wgrad_k", "kbench.py 128 --only dhead"),
        "skinny_fwd": ("k_skinny", "skinny_mfma_fwd_k<3, 128>", "kbench.py 128 --only gconv4,dconv1"),
        "skinny_wgrad": ("k_skinny", "skinny_mfma_wgrad_k<3, 128>", "kbench.py 128 --only gconv4,dconv1"),
        "config3_wino_g16_bs256": ("k_c3", "wino_gemm_g_kernel<16, 16>", "kbench.py 256 --only conv3 --pass fwd"),
        "config5_tn128x128_bs64_16to32": ("k_c5", "igemm_tng_kernel<128, 128, 2, 2>", "c5_wgrad.py")}
out = {}
for key, (tag, pat, cmd) in want.items():
    j = json.loads(subprocess.check_output([sys.executable, f"{root}/scripts/pmc_json.py", f"{root}/gpurun_out/pmc", tag, pat]))
    j["source"] = f"profiles/{src}_pmc_kernels.json: scripts/pmc_kernels.sh = separate rocprofv3 --kernel-trace --pmc passes of `python scripts/{cmd}`"
    out[key] = j
json.dump(out, open(f"{root}/gpurun_out/pmc/{src}_pmc_kernels.json", "w"), indent=1)
for k, v in out.items():
    print(k, {n: (round(v[n], 3) if isinstance(v.get(n), float) else v.get(n)) for n in ("launches_sampled", "mfma_pipe_util", "valu_per_mfma", "hbm_bytes_per_launch_corrected", "WRITE_SIZE")})
PY
