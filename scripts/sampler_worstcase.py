#!/usr/bin/env python
"""nn.BilinearSamplerBHWD backward (deterministic gather form, csrc/ops.hip) under degenerate sampling grids: a localisation net that
collapses (scale -> 0) maps every output pixel into ONE source cell.  Times the backward for scales 1, 0.5, 0.1, 0.01, 0 at the step's two
geometries and checks the gradient against a float64 scatter on the host for the collapsed grid.
python scripts/sampler_worstcase.py"""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
cg = importlib.import_module("cat-generator_amd")


def tk(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


for (N, C, H) in ((384, 64, 16), (128, 3, 32), (128, 64, 16)):
    rs = np.random.RandomState(1)
    T = lambda a: cg.Tensor(torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda().reshape(-1), a.shape)
    img, gout = T(rs.randn(N, H, H, C)), T(rs.randn(N, H, H, C))
    ys, xs = np.meshgrid(np.linspace(-1, 1, H), np.linspace(-1, 1, H), indexing="ij")
    base = np.stack([ys, xs], -1).astype(np.float32)[None].repeat(N, 0)
    m = cg.nn.BilinearSamplerBHWD()
    line = []
    for sc in (1.0, 0.5, 0.1, 0.01, 0.0):
        grid = T(base * sc + 0.013)
        m.updateOutput([img, grid])
        t = tk(lambda: m.updateGradInput([img, grid], gout))
        line.append(f"scale {sc}: {t:8.1f} us")
    print(f"N={N} C={C} {H}x{H}: " + "   ".join(line))
