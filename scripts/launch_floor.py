#!/usr/bin/env python
"""Fixed cost of one tiny dependent kernel on this GPU: 1000 chained leaky-ReLU launches on 4096 floats, eager and as
a hipGraph replay.  (The training step has ~300 such launch-bound kernels.)"""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
cg = importlib.import_module("cat-generator_amd")
lib, st = cg.tensor.lib(), cg.tensor.stream
a = torch.rand(4096, device="cuda"); b = torch.empty_like(a)
def chain(n=1000):
    for i in range(n // 2):
        lib.leakyrelu_forward(st(), a.data_ptr(), b.data_ptr(), 0.3, 4096)
        lib.leakyrelu_forward(st(), b.data_ptr(), a.data_ptr(), 0.3, 4096)
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
print("eager : %.2f us per launch" % (timeit(chain) ))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.graph(g):
    chain()
print("graph : %.2f us per launch" % (timeit(g.replay)))
