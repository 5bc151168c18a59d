#!/usr/bin/env python
"""Benchmark of the hot path: images/sec for one G+D training step (adversarial.lua:51-275) of
G32up-c / D32_st3 at 32x32 RGB, batch 128 per GPU (BASELINE.json configs[1]; N GPUs -> global batch 128*N,
configs[3] at N=8).  Synthetic data resident in HBM: real images U[0,1), noise U(-1,1), engine-generated
dropout masks, parameters initialised as weight-init.lua + Torch7 defaults.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus 8 --steps 50 --warmup 10
    python bench.py --config 3        # BASELINE configs[2]: G32up, grayscale, batch 256
    python bench.py --config 5        # per-GPU share of configs[4]: G32up-c scaled to 64x64, 64 images per GPU

Prints ONE JSON line on rank 0.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA = 157.3e12               # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_HBM = 8.0e12                       # HBM3E spec (6.29 TB/s measured with a float4 copy)
# counter passes of the roofline launches (scripts/pmc_kernels.sh), newest first
PMC_FILE = next((f for f in ("r06_pmc_kernels.json", "r05_pmc_kernels.json", "r04_pmc_kernels.json", "r03_pmc_kernels.json", "r03a_pmc_kernels.json", "r02b_pmc_kernels.json")
                 if os.path.exists(os.path.join(ROOT, "profiles", f))), "r02b_pmc_kernels.json")

CONFIGS = {   # BASELINE.json configs (index + 1)
    2: dict(name="BASELINE configs[1]: G32up-c + D32_st3, 32x32 RGB, batch 128 per GPU", gen="G32up-c", ch=3, size=32, batch=128,
            metric="images/sec per G+D step, G32up-c 32x32 RGB bs=128 per GPU"),
    3: dict(name="BASELINE configs[2]: G32up + D32_st3, 32x32 grayscale (--colorSpace=y), batch 256", gen="G32up", ch=1, size=32,
            batch=256, metric="images/sec per G+D step, G32up 32x32 grayscale bs=256 per GPU"),
    5: dict(name="per-GPU share of BASELINE configs[4]: G32up-c scaled to 64x64 RGB + D32_st3 at 64x64, 64 images per GPU "
                 "(global 512 on 8 GPUs)", gen="G32up-c", ch=3, size=64, batch=64,
            metric="images/sec per G+D step, G32up-c@64 64x64 RGB bs=64 per GPU"),
}


def step_work(cfg, N):
    """Forward FLOPs per image (2 x MAC) of the generator and the discriminator, direct count and as
    EXECUTED by the engine: a 3x3 (5x5) convolution behind nn.SpatialUpSamplingNearest(2) runs as four phase convolutions
    with 2x2 (3x3) taps on the low-res grid (16/36 resp. 36/100 of the MACs), the 5x5 layers' phases run as Winograd
    F(2x2,3x3) (16/36 of those) and the 3x3 layers' forward / data gradient as F(2x2,2x2) (9/16 of those) when the Winograd
    paths take them (planes % 128 == 0, >= 2048 tiles)."""
    C, s = cfg["ch"], cfg["size"]
    conv = lambda ci, co, k, ho: 2.0 * ci * co * k * k * ho * ho
    lin = lambda i, o: 2.0 * i * o

    def ups(ci, co, k, ho, n):
        """(direct, executed) MACs x 2 of one layer behind an upsampling; `executed` is the per-image average over the 3.5 passes a step
        makes through the generator (forward on N/2, forward on N, data gradient, weight gradient), since round 4 not the same for all."""
        d = conv(ci, co, k, ho)
        big = lambda m: ci % 128 == 0 and co % 128 == 0 and m * (ho // 2) ** 2 // 4 >= 2048      # Winograd paths: planes % 128, >= 2048 tiles
        if k == 3:      # F(2x2,2x2): forward and data gradient at 9/16 of the phase-folded MACs, the weight gradient stays phase-folded
            f = lambda m: 16 / 36 * (9 / 16 if big(m) else 1.0)
            return d, d * (0.5 * f(n // 2) + f(n) + f(n) + 16 / 36) / 3.5
        f = lambda m: 36 / 100 * (16 / 36 if big(m) else 1.0)
        return d, d * (0.5 * f(n // 2) + 3 * f(n)) / 3.5

    if cfg["gen"] == "G32up-c":
        b = s // 8
        layers = [(lin(100, 512 * b * b),) * 2, ups(512, 512, 3, 2 * b, N), ups(512, 256, 3, 4 * b, N), ups(256, 128, 5, 8 * b, N),
                  (conv(128, C, 3, 8 * b),) * 2]
    else:
        layers = [(lin(100, 128 * 64),) * 2, ups(128, 256, 5, 16, N), ups(256, 128, 5, 32, N), (conv(128, C, 3, 32),) * 2]
    fg, fg_x = sum(l[0] for l in layers), sum(l[1] for l in layers)
    loc = lambda ci, sz, p: conv(ci, 16, 3, sz // 2) + conv(16, 16, 3, sz // 2) + lin(16 * (sz // 4) ** 2, 64) + lin(64, p)
    fd = (loc(C, s, 1) + conv(C, 64, 3, s) + conv(64, 64, 3, s)
          + 3 * (loc(64, s // 2, 4) + conv(64, 64, 3, s // 2) + conv(64, 64, 3, s // 4))
          + conv(64, 128, 5, s // 2) + conv(128, 128, 7, s // 4) + lin(320 * (s // 4) ** 2, 256) + lin(256, 1))
    # What D's forward and data-gradient passes no longer issue (the executed count was overstated):
    #   * the zero-padding taps of the 7x7 layer (position-major tiles, CG_PAD_SKIP = 20 %: taken when the padding share is above it and the
    #     batch is a power of two and a multiple of the 64-row tile) - forward and data gradient; the weight gradient multiplies them still;
    #   * 20 / 36 of the 64 -> 64 plane 3x3 layers that run the fused F(2x2,3x3) kernel (csrc/wino3.hip: >= 2 workgroups per CU of 8 x 16
    #     pixel blocks; the full-resolution layer and the three 16 x 16 branch layers as one grouped launch) - forward and data gradient.
    def pad_share(k, w):
        v = sum(min(w - 1, x + k // 2) - max(0, x - k // 2) + 1 for x in range(w))
        return 1.0 - (v / (w * k)) ** 2
    skip = 0.0
    if pad_share(7, s // 4) > 0.20 and N & (N - 1) == 0 and N % 64 == 0:
        skip += pad_share(7, s // 4) * conv(128, 128, 7, s // 4)
    blocks = lambda g, sz: g * N * (sz // 8) * (sz // 16) if sz % 16 == 0 else 0
    if blocks(1, s) >= 512:
        skip += 20 / 36 * conv(64, 64, 3, s)
    if blocks(3, s // 2) >= 512:
        skip += 20 / 36 * 3 * conv(64, 64, 3, s // 2)
    #   * round 6, weight gradient: position-major K tiles (igemm_tng_kernel mode 2) leave out the padding taps of every plain layer whose
    #     padding share is >= 10 % when the batch is a multiple of 16: the 7x7 layer, the 5x5 layer at 16 x 16, the branch 3x3 layers at 8 x 8.
    skip_w = 0.0
    if N % 16 == 0:
        for k, w, macs in ((7, s // 4, conv(128, 128, 7, s // 4)), (5, s // 2, conv(64, 128, 5, s // 2)), (3, s // 4, 3 * conv(64, 64, 3, s // 4)),
                           (3, s // 2, 3 * conv(64, 64, 3, s // 2)), (3, s, conv(64, 64, 3, s))):
            if pad_share(k, w) >= 0.10:
                skip_w += pad_share(k, w) * macs
    fd_x = 4.0 * (fd - skip) + (fd - skip_w)          # two forward + two data-gradient passes and one weight-gradient pass without them
    return dict(F_G=fg, F_D=fd, W=3.5 * fg + 5.0 * fd, W_executed=3.5 * fg_x + fd_x, D_skipped_per_pass=skip, D_skipped_weight_gradient=skip_w)


def time_kernel(fn, iters=20, warm=3):
    """Average duration (s) of one launch group: `iters` launches captured into ONE hipGraph on the launch stream and the replay
    bracketed by HIP events on that stream.  (Round 2 timed eager launches from Python: the host gap between two dependent launches
    - 20-25 us of interpreter + dispatch per call - sat inside the bracket, 8 % of a 0.29 ms kernel; the replay leaves the ~2 us
    dependent-launch floor, so the figure follows the rocprofv3 duration of the same launch in profiles/r06_roofline_launch_durations.txt, a kernel trace of
    `scripts/kbench.py --only <layer>` in which every row is one of these launches and nothing else.)"""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def kernel_rooflines(cg, cfg, N):
    """The kernels that carry the step of configuration `cfg` (profiles/r06_eager_breakdown.txt for configs[1]), each timed in isolation with
    HIP events on the launch stream at the benchmarked batch: EXECUTED MFMA FLOPs per launch / duration against the fp32 MFMA
    peak.  The FIRST entry is the configuration's dominant kernel.  `traffic` / MFMA-pipe utilisation come from the committed PMC
    pass of the same launches where one exists (profiles/*_pmc_kernels.json, scripts/pmc_kernels.sh: config 2 only; bench.py
    cannot run rocprofv3 on itself) and carry their source; `traffic_over_algorithmic` prices them against STRICT algorithmic
    bytes (every input and output of the launch once)."""
    lib, stream = cg.tensor.lib(), cg.tensor.stream()
    out = []
    pmc = {}
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", PMC_FILE)))
    except Exception:
        pass
    if cfg is CONFIGS[3]:       # the counter passes of the other configurations' dominant launches carry their own keys (round 6)
        pmc = {"wino_g16": pmc.get("config3_wino_g16_bs256")} if pmc.get("config3_wino_g16_bs256") else {}
    elif cfg is CONFIGS[5]:
        pmc = {"tn128x128": pmc.get("config5_tn128x128_bs64_16to32")} if pmc.get("config5_tn128x128_bs64_16to32") else {}

    def entry(key, kernel, what, flop, t, direct=None, share=None, extra=None, alg_bytes=None):
        e = {"bound": "mfma", "kernel": kernel, "launch": what, "flop_per_launch": flop, "launch_ms": 1e3 * t,
             "achieved": flop / t / 1e12, "peak": PEAK_FP32_MFMA / 1e12, "unit": "TFLOP/s", "frac": flop / t / PEAK_FP32_MFMA,
             "traffic": None}
        if direct:
            e["flop_per_launch_direct_count"] = direct
        if share:
            e["step_time_share_profiled"] = share
        if alg_bytes:
            e["algorithmic_bytes_strict"] = alg_bytes
        p_ = pmc.get(key)
        if p_:
            e["traffic"] = p_.get("hbm_bytes_per_launch_corrected")
            e["mfma_pipe_util_pmc"] = p_.get("mfma_pipe_util")
            e["valu_per_mfma_pmc"] = p_.get("valu_per_mfma")
            e["pmc_source"] = p_.get("source", "profiles/" + PMC_FILE)
            if alg_bytes and e["traffic"]:
                e["traffic_over_algorithmic"] = e["traffic"] / alg_bytes
        if extra:
            e.update(extra)
        out.append(e)

    def conv(ci, co, k, h, n, ups):
        m = cg.nn.SpatialConvolution(ci, co, k, k, 1, 1, (k - 1) // 2)
        x = cg.Tensor(torch.rand(n * h * h * ci, device="cuda") - 0.5, (n, ci, h, h), "nhwc")
        ho = h << ups
        dy = cg.Tensor(torch.rand(n * ho * ho * co, device="cuda") - 0.5, (n, co, ho, ho), "nhwc")
        xin = cg.nn.SpatialUpSamplingNearest(2).forward(x) if ups else x
        m.forward(xin)
        return m, xin, dy, x

    E = lambda n: torch.empty(int(n), dtype=torch.float32, device="cuda")
    s = cfg["size"]

    def d_conv2():
        # D's 64->64 3x3 convolution at full resolution (models.lua:648): wino3_fused_k (csrc/wino3.hip, fused-transform F(2x2,3x3): 16/36
        # of the direct MACs) where the launch has >= 2 workgroups per CU, igemm_nn_kernel<128,64,...,16> otherwise / with CG_WINO3=0
        m2, x2, dy2, _ = conv(64, 64, 3, s, N, 0)
        f2 = 2.0 * N * s * s * 64 * 64 * 9
        strict = 4.0 * (2 * N * s * s * 64 + 576 * 64)
        import ctypes
        w3 = ctypes.c_long(0)
        lib.get_option(b"CG_WINO3", ctypes.byref(w3))
        fused = N * (s // 8) * (s // 16) >= 512 and s % 16 == 0 and w3.value != 0
        t = time_kernel(lambda: m2.updateOutput(x2))
        if fused:
            lib.set_option(b"CG_WINO3", 0)
            try:
                t_direct = time_kernel(lambda: m2.updateOutput(x2))
            finally:
                lib.set_option(b"CG_WINO3", -1)
            entry("wino3", "wino3_fused_k (wino3.hip; fused-transform Winograd F(2x2,3x3), patch in LDS, U from L2)",
                  f"updateOutput of conv3x3 64->64 @{s}x{s}, batch {N}: {N * s * s // 4} tiles x 16 positions x [64].[64 x 64]", f2 * 16 / 36, t, f2, None,
                  {"direct_kernel_ms": 1e3 * t_direct, "direct_kernel": "igemm_nn_kernel<128,64,2,2,true,true,16>",
                   "direct_kernel_frac": f2 / t_direct / PEAK_FP32_MFMA}, alg_bytes=strict + 4.0 * 16 * 64 * 64)
        else:
            entry("nn128x64", "igemm_nn_kernel<128,64,2,2,true,true,16> (gemm.hip)",
                  f"updateOutput of conv3x3 64->64 @{s}x{s}, batch {N}: M={N * s * s} K=576 N=64", f2, t, f2, None, alg_bytes=strict)

    def skinny():
        # G's last layer 128 -> C planes (models.lua:222 / :154): HBM-bound (512 bytes per pixel read once); csrc/skinny.hip
        c = cfg["ch"]
        m4, x4, dy4, _ = conv(128, c, 3, s, N, 0)
        t = time_kernel(lambda: m4.updateOutput(x4))
        tw = time_kernel(lambda: m4.accGradParameters(x4, dy4))
        by = 4.0 * N * s * s * (128 + c)
        out.append({"bound": "hbm", "kernel": "skinny_mfma_fwd_k (skinny.hip; scatter-form GEMM on the MFMA, input read once)",
                    "launch": f"updateOutput of conv3x3 128->{c} @{s}x{s}, batch {N}", "launch_ms": 1e3 * t, "achieved": by / t / 1e9,
                    "peak": PEAK_HBM / 1e9, "unit": "GB/s", "frac": by / t / PEAK_HBM, "traffic": None, "algorithmic_bytes_strict": by,
                    "wgrad_group_ms": 1e3 * tw, "wgrad_group": "skinny_mfma_wgrad_k + wgrad_reduce_small_kernel (accGradParameters)"})

    def wino5(ci, co, h, first):
        # wino_gemm_g_kernel<*,16>: the 16 Winograd-domain GEMMs + in-register output transform of an upsample2 -> conv5x5 layer
        m3, x3, dy3, _ = conv(ci, co, 5, h, N, 1)
        if not getattr(m3, "_wino", False):
            return
        y = m3.output
        v = m3._get("wino_v", (lib.conv2d_ups2_wino_v_floats(N, h, h, ci),))
        t = time_kernel(lambda: lib.conv2d_ups2_wino_gemm(cg.tensor.stream(), v.ptr, m3._u_fwd.data_ptr(), m3.bias.ptr, y.ptr, N, h, h, ci, co, 0))
        d3 = 2.0 * N * (2 * h) ** 2 * co * ci * 25
        tiles = N * h * h // 4              # 2x2-output tiles of the low-res grid: ONE V block serves the four phases
        entry("wino_g16" if first else None, "wino_gemm_g_kernel<*,16> (winograd.hip; LDS-direct loads)",
              f"forward GEMMs of upsample2 -> conv5x5 {ci}->{co} @{h}->{2 * h}, batch {N}: 4 phases x 16 GEMMs [tiles x {ci}].[{ci} x {co}] + output transform",
              d3 * 36 / 100 * 16 / 36, t, d3, None,
              {"layer_ms": {"fwd": 1e3 * time_kernel(lambda: m3.updateOutput(x3)), "dgrad": 1e3 * time_kernel(lambda: m3.updateGradInput(x3, dy3)),
                            "wgrad": 1e3 * time_kernel(lambda: m3.accGradParameters(x3, dy3))},
               "launch_input_bytes_incl_V": 4.0 * (16 * tiles * ci + 4 * 16 * ci * co + N * (2 * h) ** 2 * co),
               "note": "algorithmic_bytes_strict is the LAYER's (x + y + U): V, the transformed input this launch reads (16 x tiles x planes floats, written "
                       "by wino_input_transform_kernel = layer_ms.fwd - launch_ms), is not algorithmic; it makes one round trip through HBM / MALL"},
              alg_bytes=4.0 * (N * h * h * ci + 4 * 16 * ci * co + N * (2 * h) ** 2 * co))

    if cfg["gen"] != "G32up-c":
        # G32up (configs[2]): two upsample2 -> conv5x5 layers (models.lua:145,150), both in F(2x2,3x3); the 256->128 layer at 16->32 carries
        # 80 % of F_G, its forward GEMM launch is the configuration's dominant kernel
        wino5(256, 128, 16, True)
        wino5(128, 256, 8, False)
        d_conv2()
        skinny()
        return out

    b = s // 8            # G32up-c: 4x4 at 32x32, 8x8 at 64x64 (models.lua:196-228 scaled)
    h2 = 2 * b
    # (1) igemm_tng_kernel<128,128,2,2> (LDS-direct loads): largest direct launch = weight gradient of G's 512->256 3x3 layer behind the
    #     2x upsampling (models.lua:211-212): 4 phases x 4 taps, pixels split.  Timed: the GEMM ALONE (cg_conv2d_wgrad_gemm: partial
    #     sums left in the workspace); the launch group with its reductions rides along
    m, xin, dy, x = conv(512, 256, 3, h2, N, 1)
    geom = (N, h2, h2, 512, 256, 3, 3, 1, 1, 1)
    wsb = lib.conv2d_wgrad_workspace_bytes(*geom)
    ws = torch.empty(max(int(wsb), 16), dtype=torch.uint8, device="cuda")
    t = time_kernel(lambda: lib.conv2d_wgrad_gemm(cg.tensor.stream(), x.ptr, dy.ptr, *geom, ws.data_ptr(), wsb))
    t_group = time_kernel(lambda: m.accGradParameters(xin, dy))
    direct = 2.0 * N * (2 * h2) ** 2 * 256 * 512 * 9
    px = N * h2 * h2
    entry("tn128x128", "igemm_tng_kernel<128,128,2,2> (gemm.hip; LDS-direct loads)",
          f"weight-gradient GEMM of upsample2 -> conv3x3 512->256 @{h2}->{2 * h2}, batch {N}: 4 phases x [2048 x {px}]^T.[{px} x 256], pixels split",
          2.0 * px * 4 * 2048 * 256, t, direct, "15 % of the step's kernel time at config 2 (6 launches; profiles/r06_eager_breakdown.txt)",
          {"launch_group_ms": 1e3 * t_group, "launch_group": "accGradParameters of the layer = this GEMM + wgrad_reduce_kernel<true> + bias_part_reduce_kernel"},
          alg_bytes=4.0 * (px * 512 + 4 * px * 256 + 9 * 512 * 256))
    # (2) igemm_nng_kernel<64,128,2,2,32> (LDS-direct loads): forward of G's first convolution behind the first upsampling
    #     (models.lua:205-206), one GEMM per phase: M = N*b*b, K = 4 taps * 512, Cout = 512
    m1, x1in, dy1, _ = conv(512, 512, 3, b, N, 1)
    t = time_kernel(lambda: m1.updateOutput(x1in))
    entry("nn64x128", "igemm_nng_kernel<64,128,2,2,32> (gemm.hip; LDS-direct loads)",
          f"updateOutput of upsample2 -> conv3x3 512->512 @{b}->{2 * b}, batch {N}: 4 phases x [{N * b * b} x 2048].[2048 x 512]",
          2.0 * N * b * b * 4 * 2048 * 512, t, 2.0 * N * (2 * b) ** 2 * 512 * 512 * 9, "14 % at config 2 (15 launches)",
          alg_bytes=4.0 * (N * b * b * 512 + 4 * 2048 * 512 + N * (2 * b) ** 2 * 512))
    # (3) D's 64->64 3x3 convolution, (4) the Winograd GEMMs of G's upsample2 -> conv5x5 (models.lua:217-218)
    d_conv2()
    wino5(256, 128, 4 * b, True)
    # (5) F(2x2,2x2) on G's 512->256 3x3 layer (the layer of entry 1): forward = input transform + 4 phases x 9 GEMMs, data gradient =
    #     dy transform + 9 GEMMs over K = 4*256 in K slices + fixed-order sum - launch GROUPS through the C ABI, as the planned pass issues them
    if lib.conv2d_ups2_wino22_supported(N, h2, h2, 512, 256):
        u22, u22b = E(lib.conv2d_ups2_wino22_u_floats(512, 256)), E(lib.conv2d_ups2_wino22_u_floats(512, 256))
        lib.conv2d_ups2_wino22_pack(stream, m._wf_ph.data_ptr(), m._wb_ph.data_ptr(), u22.data_ptr(), u22b.data_ptr(), 256, 512)
        v22, vdy = E(lib.conv2d_ups2_wino22_v_floats(N, h2, h2, 512)), E(lib.conv2d_ups2_wino22_dgrad_v_floats(N, h2, h2, 512, 256))
        rows = lib.conv2d_ups2_wino_stats_rows(N, h2, h2, 512, 256)
        part, y22, g22 = E(max(int(rows), 1) * 2 * 256), E(N * (2 * h2) ** 2 * 256), E(px * 512)
        tf = time_kernel(lambda: lib.conv2d_ups2_wino22_forward_stats(cg.tensor.stream(), x.ptr, u22.data_ptr(), m.bias.ptr, y22.data_ptr(), v22.data_ptr(), N, h2, h2, 512, 256,
                                                                      part.data_ptr() if rows else None))
        tb = time_kernel(lambda: lib.conv2d_ups2_wino22_dgrad(cg.tensor.stream(), dy.ptr, u22b.data_ptr(), g22.data_ptr(), vdy.data_ptr(), N, h2, h2, 512, 256))
        f22 = 2.0 * (px // 4) * 4 * 9 * 512 * 256
        entry("wino22_fwd", "wino22_input_transform_kernel + wino_gemm_g_kernel<32,9> (winograd.hip; LDS-direct loads)",
              f"forward of upsample2 -> conv3x3 512->256 @{h2}->{2 * h2} in F(2x2,2x2), batch {N}: 4 phases x 9 GEMMs [{px // 4} tiles x 512].[512 x 256]",
              f22, tf, direct, "5 % with the data gradient at config 2 (2 + 3 launches)",
              {"timed": "launch group (transform + GEMMs)", "direct_kernel_ms": 1e3 * time_kernel(lambda: m.updateOutput(xin)),
               "dgrad_group_ms": 1e3 * tb, "dgrad_direct_kernel_ms": 1e3 * time_kernel(lambda: m.updateGradInput(xin, dy))},
              alg_bytes=4.0 * (px * 512 + 36 * 512 * 256 + 4 * px * 256))
    skinny()
    # (6) round 6: D's 7x7 layer's weight gradient on position-major K tiles (models.lua:685; igemm_tng_kernel mode 2: the 38 % of the MACs that
    #     multiply padding are not issued) and the View -> Linear head's in one kernel straight into gradWeight (models.lua:696-697; headwg.hip)
    q = s // 4
    m7, x7, dy7, _ = conv(128, 128, 7, q, N, 0)
    t7 = time_kernel(lambda: m7.accGradParameters(x7, dy7))
    d7 = 2.0 * N * q * q * 128 * 128 * 49
    valid = sum(min(q - 1, v + 3) - max(0, v - 3) + 1 for v in range(q)) ** 2 / float(q * q * 49)
    pm = N % 16 == 0 and (1.0 - valid) >= 0.10
    entry("tn128x128_d_7x7_wgrad_position_major" if pm and cfg is CONFIGS[2] else None,
          "igemm_tng_kernel<128,128,2,2> mode 2 + wgrad_reduce_small_kernel (gemm.hip; position-major K tiles)" if pm else "igemm_tng_kernel<128,128,2,2> + wgrad_reduce_small_kernel (gemm.hip)",
          f"accGradParameters of conv7x7 128->128 @{q}x{q}, batch {N}: [6272 x {N * q * q}]^T.[{N * q * q} x 128], the valid positions of every tap only",
          d7 * (valid if pm else 1.0), t7, d7, None, {"timed": "launch group (GEMM + reduction of the split partials)", "direct_count_frac": d7 / t7 / PEAK_FP32_MFMA},
          alg_bytes=4.0 * (N * q * q * 256 + 49 * 128 * 128))
    if q * q <= 64:      # the planned executor runs View -> Linear as an H x W convolution up to 64 positions (csrc/net.hip, fwd_view_gemm)
        mh = cg.nn.SpatialConvolution(320, 256, q, q, 1, 1, 0)
        xh = cg.Tensor(torch.rand(N * q * q * 320, device="cuda") - 0.5, (N, 320, q, q), "nhwc")
        dyh = cg.Tensor(torch.rand(N * 256, device="cuda") - 0.5, (N, 256, 1, 1), "nhwc")
        mh.forward(xh)
        th = time_kernel(lambda: mh.accGradParameters(xh, dyh))
        byh = 4.0 * (N * q * q * 320 + 2 * 256 * 320 * q * q + N * 256)
        e = {"bound": "hbm", "kernel": "head_wgrad_k (headwg.hip; gradWeight[co][c][tap] += dy^T x in one kernel)",
             "launch": f"accGradParameters of View -> Linear({320 * q * q}, 256), batch {N}", "launch_ms": 1e3 * th, "achieved": byh / th / 1e9,
             "peak": PEAK_HBM / 1e9, "unit": "GB/s", "frac": byh / th / PEAK_HBM, "traffic": None, "algorithmic_bytes_strict": byh,
             "mfma_frac": 2.0 * N * 320 * q * q * 256 / th / PEAK_FP32_MFMA}
        if pmc.get("head_wgrad") and cfg is CONFIGS[2]:
            e["traffic"] = pmc["head_wgrad"].get("hbm_bytes_per_launch_corrected")
            e["pmc_source"] = pmc["head_wgrad"].get("source")
        out.append(e)
    return out


def cpu_baselines(cfg):
    """CPU baselines on this box's host cores, bounded samples, timed in this run, at the METRIC's configuration.
    `value` / `kind: "port"` = the oracle (a port of the reference's algorithm class: im2col blocked over output pixels +
    register-tiled SGEMM + OpenMP over the samples of a batch, i.e. THNN SpatialConvolutionMM), at the faster of two thread counts.
    `others` carries PyTorch-CPU eager on the same graphs (oracle/torch_ref.py: oneDNN / MKL - `kind: "library"`, neither the
    reference nor the port: the best CPU library available here, usually faster than the port) and, for configs[1], the
    reference's own plumbing case configs[0] (batch 16) at its default 4 threads (train.lua:39)."""
    from oracle import oracle as O
    from oracle import torch_ref as TR
    nproc = os.cpu_count() or 1
    rs = np.random.RandomState(0)
    C, S_, NB = cfg["ch"], cfg["size"], cfg["batch"]
    mk_g = (lambda rng: O.create_G32up_c(C, 100, rng, base=S_ // 8)) if cfg["gen"] == "G32up-c" else (lambda rng: O.create_G32up(C, 100, rng))
    names = f"{cfg['gen']}/D32_st3 at {S_}x{S_}x{C}"

    def batch(n):
        return (rs.rand(n // 2, C, S_, S_).astype(np.float32), (rs.rand(n // 2, 100) * 2 - 1).astype(np.float32),
                (rs.rand(n, 100) * 2 - 1).astype(np.float32))

    def run(T, n, budget_s, max_steps):
        """one warm-up step (timed, to size the sample), then as many steps as fit the budget"""
        t0 = time.time()
        T.step(*batch(n))
        warm = time.time() - t0
        steps = int(max(1, min(max_steps, budget_s / max(warm, 1e-3))))
        t0 = time.time()
        for _ in range(steps):
            T.step(*batch(n))
        return steps, time.time() - t0

    def port(n, threads, budget_s, max_steps, note):
        O.set_num_threads(threads)
        rng = O.RNG(1)
        steps, dt = run(O.Trainer(mk_g(rng), O.create_D32_st3(C, S_, rng)), n, budget_s, max_steps)
        return {"value": n * steps / dt, "unit": "images/sec", "cores": O.num_threads(), "nproc": nproc, "kind": "port", "batch": n,
                "sample": f"{steps} G+D steps of {names} at batch {n} after 1 warm-up, oracle/ (C: im2col in L2-sized blocks + "
                          f"SGEMM, OpenMP {O.num_threads()} threads{note}), {dt:.1f} s"}

    def torch_cpu(n, threads, budget_s, max_steps):
        torch.set_num_threads(threads)
        rng = O.RNG(1)
        steps, dt = run(TR.TorchTrainer(mk_g(rng), O.create_D32_st3(C, S_, rng)), n, budget_s, max_steps)
        return {"value": n * steps / dt, "unit": "images/sec", "cores": torch.get_num_threads(), "nproc": nproc, "kind": "library", "batch": n,
                "sample": f"{steps} G+D steps of {names} at batch {n} after 1 warm-up, PyTorch-CPU eager ({torch.__version__}, autograd, "
                          f"{torch.get_num_threads()} threads) on the same graphs (oracle/torch_ref.py) - a CPU LIBRARY, not the port, "
                          f"not the reference, {dt:.1f} s"}

    wide = min(nproc, NB)                        # the port's loops are parallel over the samples of a batch
    mid = min(nproc, 32)
    head = [port(NB, wide, 8.0, 2, ", one per sample" if wide == NB else ", all cores")]
    if mid != wide:
        head.append(port(NB, mid, 8.0, 2, ""))
    others = [torch_cpu(NB, min(nproc, 128), 8.0, 2)]
    if cfg is CONFIGS[2]:
        others += [port(16, min(nproc, 4), 4.0, 3, ", configs[0] = the reference's plumbing case at its default --threads")]
    best = dict(max(head, key=lambda r: r["value"]))
    best["config"] = cfg["name"]
    best["others"] = [r for r in head if r["sample"] != best["sample"]] + others
    return best


def time_steps(cg, S, data, N, steps, warmup):
    """HIP-event time per step (ms) of `steps` eager iterations after `warmup`, on the current stream."""
    for _ in range(warmup):
        cg.adversarial.iteration(S, data, N)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        cg.adversarial.iteration(S, data, N)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


def other_config_line(cg, num, steps, warmup):
    """One of the other single-GPU configurations of BASELINE.json, timed in THIS run (VERDICT r05 #6c: the driver's command times configs[1]
    only; builder-run lines of the others are claims): ms per step, images/s, the step's executed MFMA fraction and the configuration's
    dominant kernel timed live."""
    cfg = CONFIGS[num]
    N = cfg["batch"]
    dims = (cfg["ch"], cfg["size"], cfg["size"])
    cg.manual_seed(1)
    G = cg.models.create_G(dims, 100) if cfg["gen"] == "G32up-c" else cg.models.create_G_decoder_upsampling32(dims, 100)
    D = cg.models.create_D(dims)
    S = cg.adversarial.State(dict(batchSize=N, seed=1), G, D)
    data = cg.adversarial.TrainData(np.random.RandomState(100).rand(512, *dims).astype(np.float32))
    ms = time_steps(cg, S, data, N, steps, warmup)
    w = step_work(cfg, N)
    line = {"config": num, "workload": cfg["name"], "batch": N, "steps": steps, "warmup": warmup, "ms_per_step": ms, "value": 1e3 * N / ms, "unit": "images/sec",
            "executed_frac": 1e3 * N / ms * w["W_executed"] / PEAK_FP32_MFMA, "gflop_per_image_executed": w["W_executed"] / 1e9,
            "finite": bool(np.isfinite(S.PARAMETERS_G.numpy()).all() and np.isfinite(S.PARAMETERS_D.numpy()).all())}
    del S, G, D, data
    try:
        top = kernel_rooflines(cg, cfg, N)[0]
        line["roofline"] = {k: top[k] for k in ("bound", "kernel", "launch", "launch_ms", "achieved", "peak", "unit", "frac") if k in top}
    except Exception as e:                # noqa: BLE001
        line["roofline"] = {"error": str(e)[:160]}
    torch.cuda.empty_cache()
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS), help="BASELINE.json config number (1-based)")
    ap.add_argument("--batch-per-gpu", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="(default) launch eagerly: with the planned executor a step costs the host ~1.1 ms")
    ap.add_argument("--graph", action="store_true", help="capture the iteration once (cg_graph_*) and replay the hipGraph")
    ap.add_argument("--no-kernel-roofline", action="store_true")
    ap.add_argument("--no-reference-order", action="store_true", help="skip the same-run timing of the reference's order of calls (kernel traces: the "
                    "trace tools analyse the LAST steps of the process)")
    ap.add_argument("--no-sustained", action="store_true", help="skip the 1500-step sustained-rate leg of the default one-GPU line")
    ap.add_argument("--no-dp-dry-run", action="store_true", help="skip the data-parallel dry run of the default one-GPU line")
    ap.add_argument("--dp-dry-run", type=int, nargs="?", const=8, default=None, metavar="R",
                    help="one GPU: also time the per-rank step of an R-rank data-parallel job (sync-BN, gradient buckets, D's all-reduce under the "
                         "generator forward) on single-rank RCCL communicators through cg_comm_* -> config.collectives.dry_run_ms_per_step")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the same-run lines of configs 3 and 5 (default run of config 2 on one GPU)")
    ap.add_argument("--strict-comm", action="store_true",
                    help="N > 1: the engine's transport (cg_comm_*: RCCL through the C ABI) or failure - never the torch.distributed fallback "
                         "(CG_COMM_STRICT=1).  config.collectives.transport names the transport of a run either way.")
    args = ap.parse_args()

    # stdout carries exactly ONE line, the JSON: libraries write banners there from C (RCCL's version block at communicator init,
    # gloo's "connected to N peer ranks"), so everything but the result goes to stderr from here on
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    if args.strict_comm:
        os.environ["CG_COMM_STRICT"] = "1"
    cg = importlib.import_module("cat-generator_amd")
    rank, world = cg.parallel.init_from_env()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    cg.lib()

    cfg = CONFIGS[args.config]
    N = args.batch_per_gpu or cfg["batch"]
    dims = (cfg["ch"], cfg["size"], cfg["size"])
    cg.manual_seed(1)  # identical parameters on every rank
    G = cg.models.create_G(dims, 100) if cfg["gen"] == "G32up-c" else cg.models.create_G_decoder_upsampling32(dims, 100)
    D = cg.models.create_D(dims)
    S = cg.adversarial.State(dict(batchSize=N, seed=1 + rank), G, D)
    cg.tensor.rng().offset += rank << 40  # rank-private noise / mask stream
    pool = np.random.RandomState(100 + rank).rand(1024, *dims).astype(np.float32)
    data = cg.adversarial.TrainData(pool)

    # Round 3: eager launches are the default.  With the plan below the C ABI the host needs 1.1 ms to enqueue a 7 ms step, and
    # measured on one box eager beats the hipGraph replay (7.18 vs 7.35 ms/step: the replay serialises part of the cross-stream
    # overlap); --graph keeps the capture path (cg_graph_begin / _end / _launch) measurable.
    use_graph = args.graph and not args.no_graph
    launch = "eager"
    step = lambda: cg.adversarial.iteration(S, data)
    if use_graph:
        try:
            step = cg.adversarial.GraphedIteration(S, data, N)
            launch = "hipGraph replay"
        except Exception as e:  # capture is an optimisation, not a requirement
            launch = f"eager (graph capture failed: {type(e).__name__}: {str(e)[:160]})"
            step = lambda: cg.adversarial.iteration(S, data)
    # experiment switch (profiles/r04_wgrad_stream_ab.txt): run the step on a stream of the given HIP priority (-1 = high)
    prio = os.environ.get("CG_BENCH_STREAM_PRIO")
    run_stream = torch.cuda.Stream(priority=int(prio)) if prio else torch.cuda.current_stream()
    with torch.cuda.stream(run_stream):
        for _ in range(args.warmup):
            step()
        cg.parallel.barrier()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        marks = []                     # one event per block of ~K/5 steps INSIDE the timed region (exactly K steps are timed): the spread of the line
        blk = max(1, args.steps // 5)
        for i in range(args.steps):
            step()
            if (i + 1) % blk == 0 and i + 1 < args.steps:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                marks.append((i + 1, ev))
        e1.record()
        torch.cuda.synchronize()
    cg.parallel.barrier()
    dt = time.perf_counter() - t0
    if world > 1:   # MAX over ranks on the host channel
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if torch.distributed.get_backend() == "nccl" else "cpu")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    finite = bool(np.isfinite(S.PARAMETERS_G.numpy()).all() and np.isfinite(S.PARAMETERS_D.numpy()).all())

    if rank == 0:
        ms = 1e3 * dt / args.steps
        value = N * world * args.steps / dt
        per_gpu = value / world
        w = step_work(cfg, N)
        res = {
            "metric": cfg["metric"], "value": value, "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": cfg["name"] + ", one adversarial.lua D+G update (Adam, D_L2=1e-4, clamps 1/5)",
                       "global_batch": N * world, "parallelism": f"dp{world}", "launch": launch, "executor": "cg_net_* (planned below the C ABI)" if cg.nn.planned else "per-module walk",
                       "collectives": cg.parallel.comm_info() if world > 1 else None,
                       "ms_per_sample_reference_unit": 1e3 * dt / args.steps / (N * world / 2)},
            "parity": "oracle unpinned by the reference (it holds no vectors); pinned against PyTorch-CPU autograd at operator and "
                      "whole-step level (tests/test_oracle_vs_torch.py)",
            "step_work": {"gflop_per_image_direct_count": w["W"] / 1e9, "gflop_per_image_executed": w["W_executed"] / 1e9,
                          "F_G_mflop": w["F_G"] / 1e6, "F_D_mflop": w["F_D"] / 1e6,
                          "note": "W = 3.5 F_G + 5 F_D (SURVEY.md 8d).  `executed` counts the MFMA FLOPs the engine issues after folding "
                                  "the 2x upsamplings into phase convolutions and running the 5x5 phases as Winograd F(2x2,3x3); "
                                  "`executed_frac` = executed FLOP/s / fp32 MFMA peak is the step-level MFMA utilisation, "
                                  "`direct_count_tflops` is the reference-equivalent rate (not a utilisation, may exceed the peak)",
                          "executed_tflops": per_gpu * w["W_executed"] / 1e12, "executed_frac": per_gpu * w["W_executed"] / PEAK_FP32_MFMA,
                          "direct_count_tflops": per_gpu * w["W"] / 1e12, "peak_tflops": PEAK_FP32_MFMA / 1e12},
            "event_ms_per_step": e0.elapsed_time(e1) / args.steps, "finite": finite,
        }
        if getattr(cg.lib(), "timing_probe", False):
            res["TIMING_PROBE_BUILD"] = ("the library was built with -DCG_TIMING_PROBE (halved K loops, csrc/common.h): every number of this line is a "
                                         "sensitivity probe, results are wrong by construction - NOT a measurement of the engine")
            res["value"] = None
        pts = [(0, e0)] + marks + [(args.steps, e1)]
        blocks = [pa[1].elapsed_time(pb[1]) / (pb[0] - pa[0]) for pa, pb in zip(pts[:-1], pts[1:])]
        res["spread"] = {"ms_per_step_blocks": [round(v, 4) for v in blocks], "min": min(blocks), "max": max(blocks),
                         "note": f"HIP-event time of consecutive blocks of {blk} steps inside the one timed region of {args.steps} steps"}
        # The headline numbers above are complete; the two legs below are measured live and must never cost the driver its line.
        if not args.no_kernel_roofline:
            try:
                top = kernel_rooflines(cg, cfg, N)
                res["roofline"] = dict(top[0])    # the kernel with the largest share of the step
                res["roofline_top"] = top
            except Exception as e:                # noqa: BLE001
                sw = res["step_work"]
                res["roofline"] = {"bound": "mfma", "kernel": "whole step (the live per-kernel timing failed: " + str(e)[:160] + ")",
                                   "achieved": sw["executed_tflops"], "peak": PEAK_FP32_MFMA / 1e12, "unit": "TFLOP/s",
                                   "frac": sw["executed_frac"], "traffic": None, "timed_live": False}
        # the step-level block rides inside `roofline` too: the driver's parsed line then carries the kernel AND the step
        sw = res["step_work"]
        if isinstance(res.get("roofline"), dict):
            res["roofline"]["step"] = {
                "ms_per_step": ms, "executed_tflops": sw["executed_tflops"], "executed_frac": sw["executed_frac"],
                "mfma_floor_ms": 1e3 * N * w["W_executed"] / PEAK_FP32_MFMA,     # the step's executed MFMA FLOPs at the fp32 MFMA peak
                "direct_count_over_executed": w["W"] / w["W_executed"],
                "note": "frac of the kernel block above = EXECUTED FLOPs of one launch / its duration / peak; the step's executed_frac is the "
                        "same ratio over the whole step (all launches, all phases); direct-count figures (SURVEY.md 8d's numerator) are "
                        "direct_count_over_executed times larger and are not utilisations",
                }
            if args.config == 2:
                res["roofline"]["step"].update({
                    "phases_ms_traced": {"generator forward on N/2 (fake images)": 0.92, "D forward": 0.85, "D backward + Adam": 1.13,
                                         "D forward + data gradient (G step)": 1.07, "generator backward + Adam": 1.84},
                    "phases_note": "the generator's forward on N for the G step runs BESIDE the first two phases on its own hardware queue (cg_net_forward_pair), "
                                   "so it has no interval of its own and the intervals it shares are longer than they would be alone (0.55 / 0.53 ms in the reference order)",
                    "phases_source": "profiles/r06_eager_breakdown.txt (the last of three traced steps of `rocprofv3 --kernel-trace -- python bench.py`, eager "
                                     "launches, 5.79 ms under the tracer; committed numbers, not measured in this run)"})
        # sustained rate: the same step for ~8 s (the timed region above lasts 0.1-0.3 s: a clock / thermal steady state does not show in it,
        # and a monitor that samples the chip every few seconds never sees it busy)
        if world == 1 and launch == "eager" and args.config == 2 and not args.no_sustained and not args.batch_per_gpu:
            try:
                n_sus = 1500
                blocks = [time_steps(cg, S, data, N, n_sus // 5, 0) for _ in range(5)]
                sms = sum(blocks) / len(blocks)
                res["config"]["sustained"] = {"steps": n_sus, "ms_per_step": sms, "images_per_sec": 1e3 * N / sms, "over_headline": sms / ms,
                                              "ms_per_step_blocks_of_300": blocks}
            except Exception as e:                # noqa: BLE001
                res["config"]["sustained"] = {"error": str(e)[:160]}
        # the reference's ORDER of calls (adversarial.lua:232-233 ... :185: the G-step's generator forward after D's update) in the same run:
        # what an unchanged adversarial.lua gets from the drop-in host; the headline issues the two generator forwards as one
        # cg_net_forward_pair (MODEL_G:forwardPair - six lines of adversarial.lua)
        if world == 1 and launch == "eager" and S.OPT.get("concurrent_g_both") and not args.no_reference_order:
            try:
                S.OPT["concurrent_g_both"] = False
                res["config"]["reference_order_ms"] = time_steps(cg, S, data, N, min(args.steps, 30), 3)
                res["config"]["reference_order_note"] = ("the same step with the two generator forwards issued where the reference has them "
                                                         "(CG_CONCURRENT_G_BOTH=0), same process, HIP events")
            except Exception as e:                # noqa: BLE001
                res["config"]["reference_order_ms"] = None
                res["config"]["reference_order_note"] = "failed: " + str(e)[:160]
            finally:
                S.OPT["concurrent_g_both"] = True
        if args.dp_dry_run is None:     # default: on for the plain one-GPU line of configs[1] (8 ranks = configs[3]'s plan), off otherwise
            args.dp_dry_run = 8 if (world == 1 and args.config == 2 and not args.no_dp_dry_run and not args.batch_per_gpu and launch == "eager") else 0
        if world == 1 and args.dp_dry_run > 1:
            try:
                cg.parallel.dry_run(args.dp_dry_run)
                cg.manual_seed(1)
                G2 = cg.models.create_G(dims, 100) if cfg["gen"] == "G32up-c" else cg.models.create_G_decoder_upsampling32(dims, 100)
                S2 = cg.adversarial.State(dict(batchSize=N, seed=1), G2, cg.models.create_D(dims))
                dms = time_steps(cg, S2, data, N, min(args.steps, 30), 5)
                res["config"]["collectives"] = dict(cg.parallel.comm_info(), dry_run_world=args.dp_dry_run, dry_run_ms_per_step=dms,
                                                    dry_run_over_headline=dms / ms,
                                                    note="per-rank step of the data-parallel plan on ONE GPU: sync-BN all-reduces, G's gradient buckets and "
                                                         "D's overlapped all-reduce run through cg_comm_* on single-rank RCCL communicators (device copies on the "
                                                         "communicators' streams): schedule and fork / join cost are real, xGMI traffic is absent")
                del S2, G2
            except Exception as e:                # noqa: BLE001
                res["config"]["collectives"] = {"dry_run_error": str(e)[:200]}
            finally:
                try:
                    cg.parallel.shutdown()
                except Exception:                 # noqa: BLE001
                    pass
        if world == 1 and args.config == 2 and not args.no_other_configs and not args.batch_per_gpu:
            res["other_configs"] = []
            for num in (3, 5):
                try:
                    res["other_configs"].append(other_config_line(cg, num, 20, 5))
                except Exception as e:            # noqa: BLE001
                    res["other_configs"].append({"config": num, "error": str(e)[:200]})
        if world == 1 and not args.no_cpu_baseline:
            try:
                res["cpu_baseline"] = cpu_baselines(cfg)
            except Exception as e:                # noqa: BLE001
                res["cpu_baseline"] = {"value": None, "unit": "images/sec", "cores": 0, "kind": "port", "sample": "failed: " + str(e)[:160]}
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(res) + "\n").encode())
    if world > 1:
        cg.parallel.barrier()      # rank 0 was still timing kernels: tear the communicators down together
    cg.parallel.shutdown()


if __name__ == "__main__":
    main()
