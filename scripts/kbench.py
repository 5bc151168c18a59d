#!/usr/bin/env python
"""Per-layer GEMM micro-benchmark: every conv/linear shape of G32up-c and D32_st3 at batch N (fwd, dgrad, wgrad),
timed with HIP events on the launch stream.  Usage: python scripts/kbench.py [N] [--quick] [--only a,b] [--pass fwd|dgrad|wgrad]
(--pass: launch only that pass of a convolution, so that a kernel trace / counter pass of the command holds ONE kind of launch)"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
cg = importlib.import_module("cat-generator_amd")


PASS = sys.argv[sys.argv.index("--pass") + 1] if "--pass" in sys.argv else None


def tk(fn, iters=10, warm=2):
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


def conv_case(name, N, Cin, H, Cout, k, ups, pad=None):
    pad = (k - 1) // 2 if pad is None else pad
    m = cg.nn.SpatialConvolution(Cin, Cout, k, k, 1, 1, pad)
    x = cg.Tensor(torch.rand(N * H * H * Cin, device="cuda") - 0.5, (N, Cin, H, H), "nhwc")
    Ho = (H + 2 * pad - k + 1) << ups
    dy = cg.Tensor(torch.rand(N * Ho * Ho * Cout, device="cuda") - 0.5, (N, Cout, Ho, Ho), "nhwc")
    xin = cg.nn.SpatialUpSamplingNearest(2).forward(x) if ups else x
    m.forward(xin)
    flop = 2.0 * N * Ho * Ho * Cout * Cin * k * k
    fns = {"fwd": lambda: m.updateOutput(xin), "dgrad": lambda: m.updateGradInput(xin, dy), "wgrad": lambda: m.accGradParameters(xin, dy)}
    r = {p: (tk(fns[p]) if PASS in (None, p) else float("inf")) for p in ("fwd", "dgrad", "wgrad")}
    return name, flop, r


def lin_case(name, N, i, o):
    m = cg.nn.Linear(i, o)
    x = cg.Tensor(torch.rand(N * i, device="cuda") - 0.5, (N, i))
    dy = cg.Tensor(torch.rand(N * o, device="cuda") - 0.5, (N, o))
    m.forward(x)
    flop = 2.0 * N * i * o
    r = {"fwd": tk(lambda: m.updateOutput(x)), "dgrad": tk(lambda: m.updateGradInput(x, dy)),
         "wgrad": tk(lambda: m.accGradParameters(x, dy))}
    return name, flop, r


def head_case(name, N, C, H, o):
    """View(C*H*H) -> Linear on the NHWC map the way the planned executor runs it (an H x H convolution with a 1 x 1 grid; the data
    gradient as a linear layer on the [Cout][H*W*C] operand): forward and updateGradInput through the planned net, the weight gradient
    through the convolution entry point the plan calls (cg_conv2d_wgrad with kH = H: headwg.hip)."""
    net = cg.nn.Sequential()
    net.add(cg.nn.View(C * H * H)); net.add(cg.nn.Linear(C * H * H, o))
    net.getParameters()
    x = cg.nn.as_nhwc(cg.Tensor(torch.rand(N * C * H * H, device="cuda") - 0.5, (N, C, H, H)))
    dy = cg.Tensor(torch.rand(N * o, device="cuda") - 0.5, (N, o))
    net.forward(x)
    m = cg.nn.SpatialConvolution(C, o, H, H, 1, 1, 0)
    dyc = cg.Tensor(dy.t, (N, o, 1, 1), "nhwc")
    m.forward(x)
    flop = 2.0 * N * C * H * H * o
    fns = {"fwd": lambda: net.forward(x), "dgrad": lambda: net.updateGradInput(x, dy), "wgrad": lambda: m.accGradParameters(x, dyc)}
    r = {p: (tk(fns[p]) if PASS in (None, p) else float("inf")) for p in ("fwd", "dgrad", "wgrad")}
    return name, flop, r


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 128
    quick = "--quick" in sys.argv
    only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else None
    cases = [lambda: conv_case("G.conv3 5x5 256->128 @16^2 ups", N, 256, 16, 128, 5, 1),
             lambda: conv_case("G.conv2 3x3 512->256 @8^2 ups", N, 512, 8, 256, 3, 1),
             lambda: conv_case("G.conv1 3x3 512->512 @4^2 ups", N, 512, 4, 512, 3, 1),
             lambda: conv_case("G.conv3 (N/2)", N // 2, 256, 16, 128, 5, 1),
             lambda: conv_case("G.conv2 (N/2)", N // 2, 512, 8, 256, 3, 1),
             lambda: conv_case("G.conv1 (N/2)", N // 2, 512, 4, 512, 3, 1)]
    if not quick:
        cases += [lambda: conv_case("G.conv4 3x3 128->3 @32^2", N, 128, 32, 3, 3, 0),
                  lambda: lin_case("G.linear 100->8192", N, 100, 8192),
                  lambda: conv_case("D.conv1 3x3 3->64 @32^2", N, 3, 32, 64, 3, 0),
                  lambda: conv_case("D.conv2 3x3 64->64 @32^2", N, 64, 32, 64, 3, 0),
                  lambda: conv_case("D.br conv 3x3 64->64 @16^2", N, 64, 16, 64, 3, 0),
                  lambda: conv_case("D.br conv 3x3 64->64 @8^2", N, 64, 8, 64, 3, 0),
                  lambda: conv_case("D.b4 5x5 64->128 @16^2", N, 64, 16, 128, 5, 0),
                  lambda: conv_case("D.b4 7x7 128->128 @8^2", N, 128, 8, 128, 7, 0),
                  lambda: conv_case("D.loc 3x3 64->16 @8^2", N, 64, 8, 16, 3, 0),
                  lambda: conv_case("D.loc 3x3 16->16 @16^2", N, 16, 16, 16, 3, 0),
                  # View(320*8*8) -> Linear(20480, 256) the way the planned executor runs it: an 8 x 8 convolution on the NHWC map, 1 x 1 grid
                  lambda: head_case("D.head View->Linear 20480->256", N, 320, 8, 256)]
    named = {"conv3": lambda: conv_case("G.conv3 5x5 256->128 @16^2 ups", N, 256, 16, 128, 5, 1),
             "b4": lambda: conv_case("D.b4 7x7 128->128 @8^2", N, 128, 8, 128, 7, 0),
             "b45": lambda: conv_case("D.b4 5x5 64->128 @16^2", N, 64, 16, 128, 5, 0),
             "conv1": lambda: conv_case("G.conv1 3x3 512->512 @4^2 ups", N, 512, 4, 512, 3, 1),
             "conv2": lambda: conv_case("G.conv2 3x3 512->256 @8^2 ups", N, 512, 8, 256, 3, 1),
             "dconv2": lambda: conv_case("D.conv2 3x3 64->64 @32^2", N, 64, 32, 64, 3, 0),
             "gconv4": lambda: conv_case("G.conv4 3x3 128->3 @32^2", N, 128, 32, 3, 3, 0),
             "gconv4y": lambda: conv_case("G32up conv 3x3 128->1 @32^2", N, 128, 32, 1, 3, 0),
             "dconv1": lambda: conv_case("D.conv1 3x3 3->64 @32^2", N, 3, 32, 64, 3, 0),
             "dbr16": lambda: conv_case("D.br conv 3x3 64->64 @16^2", N, 64, 16, 64, 3, 0),
             "dbr8": lambda: conv_case("D.br conv 3x3 64->64 @8^2", N, 64, 8, 64, 3, 0),
             "dlin": lambda: lin_case("D.linear 20480->256", N, 20480, 256),
             "dhead": lambda: head_case("D.head View->Linear 320x8x8->256", N, 320, 8, 256),
             "loc": lambda: conv_case("D.loc 3x3 64->16 @8^2 (3 branches stacked)", 3 * N, 64, 8, 16, 3, 0)}
    if only:
        cases = [named[o] for o in only.split(",")]
    print(f"{'layer':34s} {'GFLOP':>8s}  " + "  ".join(f"{p:>16s}" for p in ("fwd ms / TF", "dgrad ms / TF", "wgrad ms / TF")))
    tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
    for c in cases:
        name, flop, r = c()
        for p in tot:
            tot[p] += r[p]
        print(f"{name:34s} {flop / 1e9:8.2f}  " + "  ".join(f"{1e3 * r[p]:7.3f} /{flop / r[p] / 1e12:6.1f}" for p in ("fwd", "dgrad", "wgrad")), flush=True)
    print("total ms: " + "  ".join(f"{p}={1e3 * tot[p]:.3f}" for p in tot))


main()
