#!/usr/bin/env python
"""Debug/bench: Winograd path of upsample2 -> conv5x5 against the direct phase-folded path (module level)."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
cg = importlib.import_module("cat-generator_amd")

def tk(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

def run(m, xin, dy, wino):
    cg.nn.SpatialConvolution.winograd = wino
    y = m.updateOutput(xin).t.clone()
    gi = m.updateGradInput(xin, dy).t.clone()
    m.gradWeight.zero(); m.gradBias.zero()
    m.accGradParameters(xin, dy)
    res = (y, gi, m.gradWeight.t.clone(), m.gradBias.t.clone())
    t = (tk(lambda: m.updateOutput(xin)), tk(lambda: m.updateGradInput(xin, dy)), tk(lambda: m.accGradParameters(xin, dy)))
    return res, t

def case(N, Cin, H, Cout):
    m = cg.nn.SpatialConvolution(Cin, Cout, 5, 5, 1, 1, 2)
    x = cg.Tensor(torch.rand(N * H * H * Cin, device="cuda") - 0.5, (N, Cin, H, H), "nhwc")
    xin = cg.nn.SpatialUpSamplingNearest(2).forward(x)
    dy = cg.Tensor(torch.rand(N * 4 * H * H * Cout, device="cuda") - 0.5, (N, Cout, 2 * H, 2 * H), "nhwc")
    ref, td = run(m, xin, dy, False)
    got, tw = run(m, xin, dy, True)
    errs = [((a - b).abs().max() / b.abs().max()).item() for a, b in zip(got, ref)]
    print(f"N={N} {Cin}->{Cout} @{H}: rel err fwd/dgrad/wgrad/bias " + " ".join(f"{e:.1e}" for e in errs) +
          " | ms wino " + " ".join(f"{t:.3f}" for t in tw) + " direct " + " ".join(f"{t:.3f}" for t in td))

case(2, 128, 4, 128)
case(128, 256, 16, 128)
