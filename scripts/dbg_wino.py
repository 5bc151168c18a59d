#!/usr/bin/env python
"""Debug/bench: Winograd path of upsample2 -> conv5x5 against the direct phase-folded path."""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
cg = importlib.import_module("cat-generator_amd")
lib = cg.tensor.lib(); st = cg.tensor.stream()

def tk(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

def case(N, Cin, H, Cout):
    k = 5
    m = cg.nn.SpatialConvolution(Cin, Cout, k, k, 1, 1, 2)
    x = cg.Tensor(torch.rand(N * H * H * Cin, device="cuda") - 0.5, (N, Cin, H, H), "nhwc")
    xin = cg.nn.SpatialUpSamplingNearest(2).forward(x)
    dy = cg.Tensor(torch.rand(N * 4 * H * H * Cout, device="cuda") - 0.5, (N, Cout, 2 * H, 2 * H), "nhwc")
    y_ref = m.updateOutput(xin).t.clone()
    gi_ref = m.updateGradInput(xin, dy).t.clone()
    nu = lib.conv2d_ups2_wino_u_floats(Cin, Cout)
    uf = torch.empty(nu, device="cuda"); ub = torch.empty(nu, device="cuda")
    lib.conv2d_ups2_wino_pack(st, m._wf_ph.data_ptr(), m._wb_ph.data_ptr(), uf.data_ptr(), ub.data_ptr(), Cout, Cin)
    v = torch.empty(lib.conv2d_ups2_wino_v_floats(N, H, H, Cin), device="cuda")
    vdy = torch.empty(lib.conv2d_ups2_wino_v_floats(N, H, H, 4 * Cout), device="cuda")
    y = torch.empty_like(y_ref); gi = torch.empty_like(gi_ref)
    fwd = lambda: lib.conv2d_ups2_wino_forward(st, x.ptr, uf.data_ptr(), m.bias.ptr, y.data_ptr(), v.data_ptr(), N, H, H, Cin, Cout)
    bwd = lambda: lib.conv2d_ups2_wino_dgrad(st, dy.ptr, ub.data_ptr(), gi.data_ptr(), vdy.data_ptr(), N, H, H, Cin, Cout)
    fwd(); bwd()
    ef = (y - y_ref).abs().max().item() / y_ref.abs().max().item()
    eb = (gi - gi_ref).abs().max().item() / gi_ref.abs().max().item()
    t = dict(wino_fwd=tk(fwd), direct_fwd=tk(lambda: m.updateOutput(xin)), wino_dgrad=tk(bwd), direct_dgrad=tk(lambda: m.updateGradInput(xin, dy)),
             pack=tk(lambda: lib.conv2d_ups2_wino_pack(st, m._wf_ph.data_ptr(), m._wb_ph.data_ptr(), uf.data_ptr(), ub.data_ptr(), Cout, Cin)))
    print(f"N={N} {Cin}->{Cout} @{H}: rel err fwd {ef:.2e} dgrad {eb:.2e} | " + "  ".join(f"{k} {v:.3f} ms" for k, v in t.items()))

case(2, 128, 4, 128)
case(128, 256, 16, 128)
case(64, 256, 16, 128)
