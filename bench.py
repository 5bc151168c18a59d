#!/usr/bin/env python
"""Benchmark of the hot path: images/sec for one G+D training step (adversarial.lua:51-275) of
G32up-c / D32_st3 at 32x32 RGB, batch 128 per GPU (BASELINE.json configs[1]; N GPUs -> global batch 128*N,
configs[3] at N=8).  Synthetic data resident in HBM: real images U[0,1), noise U(-1,1), engine-generated
dropout masks, parameters initialised as weight-init.lua + Torch7 defaults.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus 8 --steps 20 --warmup 5

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

F_G, F_D = 2592.4e6, 374.0e6           # fwd FLOP / image
W_STEP = 3.5 * F_G + 5.0 * F_D          # necessary work per batch-image per step = 10.943 GFLOP
PEAK_FP32_MFMA = 157.3e12               # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak


def time_kernel(fn, iters=20, warm=3):
    """Average duration (s) of one launch group, HIP events on the stream the kernels are launched on."""
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def dominant_kernel_roofline(cg, N):
    """The dominant kernel is the implicit-GEMM of G's 5x5 256->128 convolution on the upsampled 32x32 map
    (models.lua:217-218; 65 % of G's FLOPs): time its forward / data-grad / weight-grad launches in isolation."""
    m = cg.nn.SpatialConvolution(256, 128, 5, 5, 1, 1, 2)
    x = cg.Tensor(torch.rand(N * 16 * 16 * 256, device="cuda") - 0.5, (N, 256, 16, 16), "nhwc")
    dy = cg.Tensor(torch.rand(N * 32 * 32 * 128, device="cuda") - 0.5, (N, 128, 32, 32), "nhwc")
    xin = cg.nn.SpatialUpSamplingNearest(2).forward(x)
    m.forward(xin)
    flop = 2.0 * N * 32 * 32 * 128 * 256 * 25
    out = {}
    out["igemm_nn_fwd"] = time_kernel(lambda: m.updateOutput(xin))
    out["igemm_nn_dgrad"] = time_kernel(lambda: m.updateGradInput(xin, dy))
    out["igemm_tn_wgrad"] = time_kernel(lambda: m.accGradParameters(xin, dy))
    return flop, out


def cpu_baseline(steps=16, N=16):
    """The oracle (a port: im2col + blocked SGEMM + OpenMP, the algorithm class of THNN SpatialConvolutionMM)
    timed on this box's host cores on a bounded sample: `steps` iterations at batch 16 (BASELINE configs[0])."""
    from oracle import oracle as O
    # the port parallelises its convolutions over the N samples of the batch: more threads than that only spin
    O.set_num_threads(min(os.cpu_count() or 1, N))
    rng = O.RNG(1)
    T = O.Trainer(O.create_G32up_c(3, 100, rng), O.create_D32_st3(3, 32, rng))
    rs = np.random.RandomState(0)

    def one():
        real = rs.rand(N // 2, 3, 32, 32).astype(np.float32)
        nd = (rs.rand(N // 2, 100) * 2 - 1).astype(np.float32)
        ng = (rs.rand(N, 100) * 2 - 1).astype(np.float32)
        T.step(real, nd, ng)

    one()
    t0 = time.time()
    for _ in range(steps):
        one()
    dt = time.time() - t0
    return {"value": N * steps / dt, "unit": "images/sec", "cores": O.num_threads(), "kind": "port",
            "sample": f"{steps} G+D steps of G32up-c/D32_st3 at batch {N} (configs[0]) after 1 warm-up, "
                      f"oracle/ (C im2col+SGEMM, OpenMP {O.num_threads()} threads), {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch-per-gpu", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying a hipGraph")
    ap.add_argument("--graph", action="store_true", help="force hipGraph replay also for N > 1")
    ap.add_argument("--no-kernel-roofline", action="store_true")
    args = ap.parse_args()

    cg = importlib.import_module("cat-generator_amd")
    rank, world = cg.parallel.init_from_env()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    cg.lib()

    N = args.batch_per_gpu
    cg.manual_seed(1)  # identical parameters on every rank
    G = cg.models.create_G((3, 32, 32), 100)
    D = cg.models.create_D((3, 32, 32))
    S = cg.adversarial.State(dict(batchSize=N, seed=1 + rank), G, D)
    cg.tensor.rng().offset += rank << 40  # rank-private noise / mask stream
    pool = np.random.RandomState(100 + rank).rand(1024, 3, 32, 32).astype(np.float32)
    data = cg.adversarial.TrainData(pool)

    use_graph = (world == 1 and not args.no_graph) or args.graph
    launch = "eager"
    step = lambda: cg.adversarial.iteration(S, data)
    if use_graph:
        try:
            step = cg.adversarial.GraphedIteration(S, data, N)
            launch = "hipGraph replay"
        except Exception as e:  # capture is an optimisation, not a requirement
            launch = f"eager (graph capture failed: {type(e).__name__})"
            step = lambda: cg.adversarial.iteration(S, data)
    for _ in range(args.warmup):
        step()
    cg.parallel.barrier()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        step()
    e1.record()
    torch.cuda.synchronize()
    cg.parallel.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    finite = bool(np.isfinite(S.PARAMETERS_G.numpy()).all() and np.isfinite(S.PARAMETERS_D.numpy()).all())

    if rank == 0:
        ms = 1e3 * dt / args.steps
        value = N * world * args.steps / dt
        per_gpu = value / world
        res = {
            "metric": "images/sec per G+D step, G32up-c 32x32 RGB bs=128 per GPU",
            "value": value, "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: G32up-c + D32_st3, 32x32 RGB, batch 128 per GPU, one "
                                   "adversarial.lua D+G update (Adam, D_L2=1e-4, clamps 1/5)",
                       "global_batch": N * world, "parallelism": f"dp{world}", "launch": launch,
                       "ms_per_sample_reference_unit": 1e3 * dt / args.steps / (N * world / 2)},
            "step_roofline": {"bound": "mfma", "work_gflop_per_image": W_STEP / 1e9,
                              "note": "direct-count necessary work W = 3.5 F_G + 5 F_D (SURVEY.md 8d); the engine "
                                      "executes ~5.42 GFLOP/image after folding the upsamplings",
                              "achieved": per_gpu * W_STEP / 1e12, "peak": PEAK_FP32_MFMA / 1e12,
                              "unit": "TFLOP/s", "frac": per_gpu * W_STEP / PEAK_FP32_MFMA},
            "event_ms_per_step": e0.elapsed_time(e1) / args.steps, "finite": finite,
        }
        if not args.no_kernel_roofline:
            flop, t = dominant_kernel_roofline(cg, N)
            # HBM traffic of that launch from the PMC pass committed under profiles/ (2*FETCH_SIZE + WRITE_SIZE,
            # the gfx950 correction of MI355X_MICROARCH.md); bench.py cannot run rocprofv3 on itself
            traffic, util = None, None
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", "r01c_pmc_dominant_conv.json")))
                k = [v for name, v in pmc.items() if "igemm_nn" in name and "grid=262144" in name][0]
                traffic, util = k["hbm_bytes_per_launch_corrected"], k["mfma_pipe_util"]
            except Exception:
                pass
            # executed FLOPs: the 4-phase form of upsample2 -> conv5x5 runs 4 x 3x3 taps per low-res pixel
            flop_exec = flop * 36.0 / 100.0
            res["roofline"] = {"bound": "mfma", "kernel": "igemm_nn_kernel<128,128,2,2,FAST> (G conv5x5 256->128 @32x32 "
                               "with the 2x nearest upsampling folded in as 4 phase convs, batch %d)" % N,
                               "achieved": flop_exec / t["igemm_nn_fwd"] / 1e12, "peak": PEAK_FP32_MFMA / 1e12,
                               "unit": "TFLOP/s", "frac": flop_exec / t["igemm_nn_fwd"] / PEAK_FP32_MFMA,
                               "traffic": traffic, "mfma_pipe_util_pmc": util,
                               "note": "achieved/frac count EXECUTED MFMA FLOPs (2.78x fewer than the direct-count "
                                       "algorithmic FLOPs of the layer); direct-count rate in `tflops_direct`",
                               "flop_per_launch": flop_exec, "flop_per_launch_direct": flop,
                               "launch_ms": {k: 1e3 * v for k, v in t.items()},
                               "tflops_direct": {k: flop / v / 1e12 for k, v in t.items()},
                               "tflops_executed": {k: flop_exec / v / 1e12 for k, v in t.items()}}
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline()
        print(json.dumps(res), flush=True)
    cg.parallel.shutdown()


if __name__ == "__main__":
    main()
