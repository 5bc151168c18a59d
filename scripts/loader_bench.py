#!/usr/bin/env python
"""Input-pipeline cost per epoch pool (dataset.AsyncLoader's device side): pinned 8-bit buffer -> cg_memcpy_h2d on a copy stream ->
cg_images_u8_to_f32, against uploading the same images as pageable fp32 on the training stream.  Usage: loader_bench.py [N] [S]"""
import ctypes, importlib, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
cg = importlib.import_module("cat-generator_amd")
L = cg.lib()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 32
nbytes = N * S * S * 3
host, dev, cs, ev = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
L.host_alloc(ctypes.byref(host), nbytes); L.malloc(ctypes.byref(dev), nbytes); L.stream_create(ctypes.byref(cs)); L.event_create(ctypes.byref(ev))
np.ctypeslib.as_array(ctypes.cast(host, ctypes.POINTER(ctypes.c_uint8)), shape=(nbytes,))[:] = np.random.RandomState(0).randint(0, 256, nbytes, dtype=np.uint8)
pool = cg.Tensor.empty((N, 3, S, S), "nhwc")


def pinned():
    L.memcpy_h2d(cs, dev, host, nbytes); L.images_u8_to_f32(cs, dev, pool.ptr, N * S * S, 0); L.event_record(ev, cs); L.event_sync(ev)


f32 = np.random.RandomState(1).rand(N, 3, S, S).astype(np.float32)


def pageable():
    cg.adversarial.TrainData(f32); torch.cuda.synchronize()


for name, fn in (("pinned u8 + copy stream + device conversion", pinned), ("pageable fp32 upload + layout kernel", pageable)):
    for _ in range(3):
        fn()
    t0 = time.perf_counter()
    for _ in range(20):
        fn()
    dt = (time.perf_counter() - t0) / 20
    print(f"{name}: {1e3 * dt:.3f} ms per pool of {N} images {S}x{S} ({N / dt / 1e6:.2f} M images/s)")
