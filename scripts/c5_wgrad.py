#!/usr/bin/env python
"""The dominant launch of config 5 (weight gradient of upsample2 -> conv3x3 512->256 at 16 -> 32, batch 64) alone, for a PMC pass."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cg = importlib.import_module("cat-generator_amd")
N, h = 64, 16
m = cg.nn.SpatialConvolution(512, 256, 3, 3, 1, 1, 1)
x = cg.Tensor(torch.rand(N * h * h * 512, device="cuda") - 0.5, (N, 512, h, h), "nhwc")
dy = cg.Tensor(torch.rand(N * 4 * h * h * 256, device="cuda") - 0.5, (N, 256, 2 * h, 2 * h), "nhwc")
xin = cg.nn.SpatialUpSamplingNearest(2).forward(x)
m.forward(xin)
for _ in range(12):
    m.accGradParameters(xin, dy)
torch.cuda.synchronize()
print("done")
