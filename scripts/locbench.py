#!/usr/bin/env python
"""The localisation launches alone (cg_locnet_forward / _backward, csrc/locnet.hip) at batch N: the first transformer (1 group, S 16,
3 planes) and the three branch transformers (3 groups on a shared [N,16,16,64] map, S 8), timed with HIP events around `iters`
back-to-back launches.
    python scripts/locbench.py [N] [iters]"""
import ctypes
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
cg = importlib.import_module("cat-generator_amd")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
L, st = cg.lib(), cg.tensor.stream()
dev = "cuda"


def case(name, G, S, Cin, P, ur, us, ut, shared):
    rs = np.random.RandomState(S)
    K3 = 16 * (S // 2) ** 2
    T = lambda *shape, sc=1.0: torch.from_numpy((rs.randn(*shape) * sc).astype(np.float32)).to(dev)
    keep, ptrs = [], []
    for g in range(G):
        w1, b1, w2, b2 = T(16, Cin, 3, 3, sc=0.1), T(16, sc=0.1), T(16, 16, 3, 3, sc=0.1), T(16, sc=0.1)
        w3, b3, w4, b4 = T(64, K3, sc=0.05), T(64, sc=0.1), T(P, 64, sc=0.05), T(P, sc=0.1)
        wf1, wb1, wf2, wb2 = (torch.empty(9 * Cin * 16, device=dev), torch.empty(9 * 16 * Cin, device=dev), torch.empty(9 * 256, device=dev),
                              torch.empty(9 * 256, device=dev))
        L.pack_conv_weight(st, w1.data_ptr(), wf1.data_ptr(), wb1.data_ptr(), 16, Cin, 3, 3)
        L.pack_conv_weight(st, w2.data_ptr(), wf2.data_ptr(), wb2.data_ptr(), 16, 16, 3, 3)
        ts = [w1, b1, w2, b2, w3, b3, w4, b4, wf1, wb1, wf2, wb2]
        keep += ts
        ptrs += [t.data_ptr() for t in ts]
    W = (ctypes.c_void_p * len(ptrs))(*ptrs)
    GN = G * N
    x = T(N if shared else GN, 2 * S, 2 * S, Cin)
    E = lambda *shape: torch.empty(*shape, device=dev)
    pooled, h1, m2, h2, h3, prm, grid = E(GN, S, S, Cin), E(GN, S, S, 16), E(GN, S, S, 16), E(GN, K3), E(GN, 64), E(GN, P), E(GN, 2 * S, 2 * S, 2)
    ggrid = T(GN, 2 * S, 2 * S, 2)
    ga1, ga2, g3, g4, gx = E(GN, S, S, 16), E(GN, S, S, 16), E(GN, 64), E(GN, P), E(GN, 2 * S, 2 * S, Cin)
    f = lambda: L.locnet_forward(st, G, N, x.data_ptr(), 1 if shared else 0, W, S, Cin, P, ur, us, ut, 0.333, 2 * S, 2 * S, pooled.data_ptr(), h1.data_ptr(),
                                 m2.data_ptr(), h2.data_ptr(), h3.data_ptr(), prm.data_ptr(), grid.data_ptr())
    b = lambda: L.locnet_backward(st, G, N, W, S, Cin, P, ur, us, ut, 0.333, 2 * S, 2 * S, h1.data_ptr(), m2.data_ptr(), h3.data_ptr(), prm.data_ptr(),
                                  ggrid.data_ptr(), ga1.data_ptr(), ga2.data_ptr(), g3.data_ptr(), g4.data_ptr(), gx.data_ptr())
    out = []
    for fn in (f, b):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / iters * 1e3)
    chk = float(grid.double().abs().sum() + gx.double().abs().sum())
    print(f"{name}: forward {out[0]:7.1f} us  backward {out[1]:7.1f} us   (G {G} x N {N}, S {S}, Cin {Cin}; checksum {chk:.6e})")


case("first transformer  ", 1, 16, 3, 1, 1, 0, 0, False)
case("branch transformers", 3, 8, 64, 4, 1, 1, 1, True)
