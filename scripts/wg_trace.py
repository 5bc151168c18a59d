#!/usr/bin/env python
"""Per-workgroup timeline of igemm_nn_kernel launches (library built with CG_BUILD_DEFINES=-DCG_TRACE, cat-generator_amd/build.py): dispatch balance over
the CUs and the time every workgroup spends in its prologue, K loop and epilogue.
Usage: CATGAN_LIB=$PWD/cat-generator_amd/lib/libcatgan_hip_exptrace.so python scripts/wg_trace.py"""
import collections
import ctypes
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
cg = importlib.import_module("cat-generator_amd")
cg.lib()
dbg = ctypes.CDLL(os.environ["CATGAN_LIB"])
MAXWG = 1 << 16
buf = torch.zeros(MAXWG * 8, dtype=torch.int64, device="cuda")
assert dbg.cg_debug_set_trace(ctypes.c_void_p(buf.data_ptr())) == 0


def report(tag, fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    buf.zero_()
    torch.cuda.synchronize()
    fn()
    torch.cuda.synchronize()
    t = buf.cpu().numpy().reshape(MAXWG, 8)
    t = t[t[:, 0] != 0]
    t0 = t[:, 0].min()
    st, pro, loop, epi = (t[:, 0] - t0) / 100.0, (t[:, 1] - t[:, 0]) / 100.0, (t[:, 2] - t[:, 1]) / 100.0, (t[:, 3] - t[:, 2]) / 100.0
    end = (t[:, 3] - t0) / 100.0
    cu = (t[:, 5] & 0xf) * 65536 + ((t[:, 4] >> 8) & 0xff)
    per_cu = collections.Counter(cu.tolist())
    hist = collections.Counter(per_cu.values())
    q = lambda a: f"min {a.min():7.1f}  p50 {np.median(a):7.1f}  p90 {np.quantile(a, 0.9):7.1f}  max {a.max():7.1f}"
    print(f"== {tag}: {len(t)} workgroups on {len(per_cu)} CUs; workgroups per CU: " + ", ".join(f"{k}: {v} CUs" for k, v in sorted(hist.items())))
    print(f"   span (first start .. last end) {end.max():.1f} us")
    print(f"   start offset [us]   {q(st)}")
    print(f"   prologue     [us]   {q(pro)}")
    if (t[:, 6] != 0).all():     # igemm_nn_kernel also stamps the end of its integer set-up
        print(f"     set-up     [us]   {q((t[:, 6] - t[:, 0]) / 100.0)}")
        print(f"     1st tile   [us]   {q((t[:, 1] - t[:, 6]) / 100.0)}")
    print(f"   K loop       [us]   {q(loop)}")
    print(f"   epilogue     [us]   {q(epi)}")
    print(f"   end time     [us]   {q(end)}")
    late = st > 5.0
    if late.any():
        print(f"   {int(late.sum())} workgroups start later than 5 us (second round on a busy CU): their K loop {q(loop[late])}")
        print(f"   the others' K loop {q(loop[~late])}")


def conv(N, Cin, H, Cout, k, ups):
    m = cg.nn.SpatialConvolution(Cin, Cout, k, k, 1, 1, (k - 1) // 2)
    x = cg.Tensor(torch.rand(N * H * H * Cin, device="cuda") - 0.5, (N, Cin, H, H), "nhwc")
    Ho = H << ups
    dy = cg.Tensor(torch.rand(N * Ho * Ho * Cout, device="cuda") - 0.5, (N, Cout, Ho, Ho), "nhwc")
    xin = cg.nn.SpatialUpSamplingNearest(2).forward(x) if ups else x
    m.forward(xin)
    return m, xin, dy


N = 128
if "--dconv2" in sys.argv:
    m2, x2, dy2 = conv(N, 64, 32, 64, 3, 0)
    report("D.conv2 64->64 @32 forward (igemm_nn<128,64,..,16>)", lambda: m2.updateOutput(x2))
    report("D.conv2 64->64 @32 data gradient", lambda: m2.updateGradInput(x2, dy2))
    sys.exit(0)
m, xin, dy = conv(N, 512, 8, 256, 3, 1)
report("G.conv2 512->256 @8->16 dgrad (igemm_nn<64,128,..,32>, M 8192 K 4096 N 512)", lambda: m.updateGradInput(xin, dy))
report("G.conv2 forward (4 phases)", lambda: m.updateOutput(xin))
m2, x2, dy2 = conv(N, 64, 32, 64, 3, 0)
report("D.conv2 64->64 @32 forward (igemm_nn<128,64,..,16>)", lambda: m2.updateOutput(x2))
m3, x3, dy3 = conv(N, 128, 8, 128, 7, 0)
report("D.b4 7x7 128->128 @8 forward", lambda: m3.updateOutput(x3))
m4, x4, dy4 = conv(N, 512, 4, 512, 3, 1)
report("G.conv1 512->512 @4->8 dgrad (split-K)", lambda: m4.updateGradInput(x4, dy4))
