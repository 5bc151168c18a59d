#!/usr/bin/env python
"""BilinearSamplerBHWD backward: deterministic gather kernel vs the atomic scatter (CG_SAMPLER_ATOMICS), HIP-event timed."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
cg = importlib.import_module("cat-generator_amd")
L, st = cg.lib(), cg.tensor.stream()


def tk(fn, iters=30, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


for N, C, H in ((128, 64, 16), (384, 64, 16), (128, 3, 32), (64, 3, 64)):
    img = torch.rand(N * H * H * C, device="cuda"); gout = torch.rand(N * H * H * C, device="cuda")
    th = (torch.rand(N, device="cuda") - 0.5) * 0.6
    ys = torch.linspace(-1, 1, H, device="cuda")
    gy, gx = torch.meshgrid(ys, ys, indexing="ij")
    c, s = torch.cos(th)[:, None, None], torch.sin(th)[:, None, None]
    grid = torch.stack([c * gy - s * gx, s * gy + c * gx], dim=-1).contiguous()
    gimg = torch.empty_like(img); ggrid = torch.empty_like(grid)
    res = {}
    for mode in (0, 1):
        L.set_option(b"CG_SAMPLER_ATOMICS", mode)
        f = lambda: L.bilinear_sampler_backward(st, img.data_ptr(), grid.data_ptr(), gout.data_ptr(), gimg.data_ptr(), ggrid.data_ptr(), N, H, H, C, H, H)
        res[mode] = tk(f)
        if mode == 0:
            ref = (gimg.clone(), ggrid.clone())
    err = float((gimg - ref[0]).abs().max()), float((ggrid - ref[1]).abs().max())
    L.set_option(b"CG_SAMPLER_ATOMICS", -1)
    print(f"N={N} C={C} {H}x{H}: deterministic {res[0]:.1f} us, atomics {res[1]:.1f} us, max|d gimg|={err[0]:.2e} max|d ggrid|={err[1]:.2e}")
