#!/usr/bin/env python
"""The forward of nn.SpatialUpSamplingNearest(2) -> 3x3 convolution alone (G's 512 -> 256 layer, models.lua:211-212) at batch N: the
phase-folded direct kernels (cg_conv2d_forward with ups = 1, cg_conv2d_dgrad_ups2) beside F(2x2,2x2) (cg_conv2d_ups2_wino22_*, csrc/winograd.hip),
HIP events around `iters` back-to-back launches, plus the largest difference of the two outputs.
    python scripts/wino22_bench.py [N] [Cin] [Cout] [Hp] [iters]"""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
cg = importlib.import_module("cat-generator_amd")
a = [int(v) for v in sys.argv[1:]]
N, Cin, Cout, H, iters = (a + [128, 512, 256, 8, 50][len(a):])[:5]
L, st = cg.lib(), cg.tensor.stream()
dev = "cuda"
rs = np.random.RandomState(1)
T = lambda *shape, sc=1.0: torch.from_numpy((rs.randn(*shape) * sc).astype(np.float32)).to(dev)
E = lambda n: torch.empty(int(n), dtype=torch.float32, device=dev)
w, bias, x = T(Cout, Cin, 3, 3, sc=1 / np.sqrt(Cin * 9)), T(Cout), T(N, H, H, Cin)
n_ph = L.pack_conv_weight_ups2_floats(Cout, Cin, 3, 1)
wf, wb = E(n_ph), E(n_ph)
L.pack_conv_weight_ups2(st, w.data_ptr(), wf.data_ptr(), wb.data_ptr(), Cout, Cin, 3, 1)
u22, u22b, v = E(L.conv2d_ups2_wino22_u_floats(Cin, Cout)), E(L.conv2d_ups2_wino22_u_floats(Cin, Cout)), E(L.conv2d_ups2_wino22_v_floats(N, H, H, Cin))
L.conv2d_ups2_wino22_pack(st, wf.data_ptr(), wb.data_ptr(), u22.data_ptr(), u22b.data_ptr(), Cout, Cin)
dy, vdy = T(N, 2 * H, 2 * H, Cout), E(L.conv2d_ups2_wino22_dgrad_v_floats(N, H, H, Cin, Cout))
g0, g1 = E(N * H * H * Cin), E(N * H * H * Cin)
dwsb = L.conv2d_dgrad_ups2_workspace_bytes(N, H, H, Cin, Cout, 3, 1)
dws = torch.empty(max(int(dwsb), 16), dtype=torch.uint8, device=dev)
y0, y1 = E(N * 4 * H * H * Cout), E(N * 4 * H * H * Cout)
geom = (N, H, H, Cin, Cout, 3, 3, 1, 1, 1)
wsb = L.conv2d_workspace_bytes(*geom)
ws = torch.empty(max(int(wsb), 16), dtype=torch.uint8, device=dev)
rows = L.conv2d_ups2_wino_stats_rows(N, H, H, Cin, Cout)
part = torch.zeros(max(int(rows), 1) * 2 * Cout, dtype=torch.float32, device=dev)
direct = lambda: L.conv2d_forward(st, x.data_ptr(), wf.data_ptr(), bias.data_ptr(), y0.data_ptr(), *geom, ws.data_ptr(), wsb)
wino = lambda: L.conv2d_ups2_wino22_forward_stats(st, x.data_ptr(), u22.data_ptr(), bias.data_ptr(), y1.data_ptr(), v.data_ptr(), N, H, H, Cin,
                                                  Cout, part.data_ptr() if rows else None)
gw0, gw1, gb0 = E(Cout * Cin * 9).zero_(), E(Cout * Cin * 9).zero_(), E(Cout).zero_()
wwsb = L.conv2d_wgrad_workspace_bytes(*geom)
wws = torch.empty(max(int(wwsb), 16), dtype=torch.uint8, device=dev)
wdirect = lambda: L.conv2d_wgrad(st, x.data_ptr(), dy.data_ptr(), gw0.data_ptr(), gb0.data_ptr(), *geom, 1.0, wws.data_ptr(), wwsb)
ddirect = lambda: L.conv2d_dgrad_ups2(st, dy.data_ptr(), wb.data_ptr(), g0.data_ptr(), N, H, H, Cin, Cout, 3, 1, dws.data_ptr(), dwsb)
dwino = lambda: L.conv2d_ups2_wino22_dgrad(st, dy.data_ptr(), u22b.data_ptr(), g1.data_ptr(), vdy.data_ptr(), N, H, H, Cin, Cout)
for name, fn in (("forward, direct (4 phases x 2x2 taps)", direct), ("forward, F(2x2,2x2)", wino), ("data gradient, direct", ddirect),
                 ("data gradient, F(2x2,2x2)", dwino), ("weight gradient (+ bias), direct", wdirect)):
    for _ in range(3):
        assert fn() == 0
    if fn is wdirect: gw0.zero_()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:38s} {e0.elapsed_time(e1) / iters * 1e3:8.1f} us   (N {N}, {H}x{H} -> {2 * H}x{2 * H}, {Cin} -> {Cout})")
print("forward      : max |direct - F(2x2,2x2)| = %.3e (max |y| %.3e)" % (float((y0 - y1).abs().max()), float(y0.abs().max())))
print("data gradient: max |direct - F(2x2,2x2)| = %.3e (max |g| %.3e)" % (float((g0 - g1).abs().max()), float(g0.abs().max())))
