#!/usr/bin/env python
"""Debug: cg_pack_conv_weight / _ups2 against numpy for every layer shape of G and D."""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
cg = importlib.import_module("cat-generator_amd")
lib = cg.tensor.lib()
st = cg.tensor.stream()

def pm(a, d, pad): return ((a + d - pad) >> 1) - ((a - pad) >> 1)

def plain(Cout, Cin, k):
    KK = k * k
    w = torch.rand(Cout * Cin * KK, device="cuda")
    wf = torch.zeros_like(w); wb = torch.zeros_like(w)
    lib.pack_conv_weight(st, w.data_ptr(), wf.data_ptr(), wb.data_ptr(), Cout, Cin, k, k)
    W = w.cpu().numpy().reshape(Cout, Cin, KK)
    ef = W.transpose(2, 1, 0).reshape(-1)
    eb = W[:, :, ::-1].transpose(2, 0, 1).reshape(-1)
    print("plain", Cout, Cin, k, "wf", np.abs(wf.cpu().numpy() - ef).max(), "wb", np.abs(wb.cpu().numpy() - eb).max())

def ups(Cout, Cin, k):
    pad = (k - 1) // 2
    kp = 2 if k == 3 else 3
    n = lib.pack_conv_weight_ups2_floats(Cout, Cin, k, pad)
    w = torch.rand(Cout * Cin * k * k, device="cuda")
    wf = torch.zeros(n, device="cuda"); wb = torch.zeros(n, device="cuda")
    lib.pack_conv_weight_ups2(st, w.data_ptr(), wf.data_ptr(), wb.data_ptr(), Cout, Cin, k, pad)
    W = w.cpu().numpy().reshape(Cout, Cin, k, k).astype(np.float64)
    E = np.zeros((4, kp, kp, Cin, Cout))
    for p in range(4):
        for dy in range(k):
            for dx in range(k):
                E[p, pm(p >> 1, dy, pad), pm(p & 1, dx, pad)] += W[:, :, dy, dx].T
    ef = E.reshape(-1); eb = E.transpose(0, 1, 2, 4, 3).reshape(-1)
    print("ups", Cout, Cin, k, "wf", np.abs(wf.cpu().numpy() - ef).max(), "wb", np.abs(wb.cpu().numpy() - eb).max())

for c in [(8192, 100, 1), (512, 512, 3), (256, 512, 3), (128, 256, 5), (3, 128, 3), (64, 3, 3), (64, 64, 3), (128, 64, 5),
          (128, 128, 7), (16, 64, 3), (16, 16, 3), (256, 20480, 1), (1, 256, 1), (4, 64, 1), (20, 50, 5), (5, 3, 3)]:
    plain(*c)
for c in [(512, 512, 3), (256, 512, 3), (128, 256, 5), (5, 3, 3), (33, 20, 5)]:
    ups(*c)
