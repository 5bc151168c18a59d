#!/usr/bin/env python
"""D32_st3 planned forward + backward in a loop (for rocprofv3 --kernel-trace --stats): python scripts/dbench.py [N] [iters]"""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
cg = importlib.import_module("cat-generator_amd")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
cg.manual_seed(1)
D = cg.models.create_D((3, 32, 32))
p, g = D.getParameters()
p.copy(p.numpy() + (np.random.RandomState(0).randn(p.nElement()) * 0.01).astype(np.float32))
x = cg.nn.as_nhwc(cg.Tensor.from_numpy(np.random.RandomState(1).rand(N, 3, 32, 32).astype(np.float32)))
dy = cg.Tensor.from_numpy(np.random.RandomState(2).randn(N, 1).astype(np.float32))
for _ in range(3):
    D.forward(x); D.backward(x, dy)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    D.forward(x); D.backward(x, dy)
e1.record(); torch.cuda.synchronize()
print(f"D fwd+bwd N={N}: {e0.elapsed_time(e1) / iters:.3f} ms")
