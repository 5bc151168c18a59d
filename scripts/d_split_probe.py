#!/usr/bin/env python
"""Probe (round 5): is D's pass over N rows faster as TWO passes over N/2 rows on two hardware queues?  D32_st3 has no batch statistics,
so a row split changes nothing but the order of the weight-gradient sums.  The passes of D are chain-bound (one wave of workgroups per
launch, all in their prologue / epilogue at the same time): a second chain on another queue could fill those ends.
Usage: python scripts/d_split_probe.py [N]   -> ms per forward+backward: one pass at N | two passes at N/2 side by side | two passes at N/2 in line"""
import ctypes
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
cg = importlib.import_module("cat-generator_amd")
nn = cg.nn


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    cg.manual_seed(3)
    nn.planned = True
    nets = [cg.models.create_D((3, 32, 32)) for _ in range(3)]
    for D in nets:
        D.training()
    xs = [cg.Tensor(torch.rand(n * 3 * 32 * 32, device="cuda"), (n, 3, 32, 32), "nhwc") for n in (N, N // 2, N // 2)]
    gys = [cg.Tensor(torch.rand(n, device="cuda") - 0.5, (n, 1)) for n in (N, N // 2, N // 2)]
    h = ctypes.c_void_p()
    cg.lib().stream_on_queue(cg.tensor.stream(), 2, 7, ctypes.byref(h))
    side = torch.cuda.ExternalStream(h.value)
    e_f, e_j = torch.cuda.Event(), torch.cuda.Event()

    def one(i):
        nets[i].forward(xs[i]); nets[i].backward(xs[i], gys[i])

    def full():
        one(0)

    def split_side():
        e_f.record(); side.wait_event(e_f)
        with torch.cuda.stream(side):
            one(2); e_j.record()
        one(1)
        torch.cuda.current_stream().wait_event(e_j)

    def split_line():
        one(1); one(2)

    def tk(fn, iters=30, warm=5):
        for _ in range(warm):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); a.record()
        for _ in range(iters):
            fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / iters

    for rep in range(2):
        print(f"N={N}: one pass {tk(full):.3f} ms | two half passes side by side {tk(split_side):.3f} ms | in line {tk(split_line):.3f} ms", flush=True)


if __name__ == "__main__":
    main()
