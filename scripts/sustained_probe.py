#!/usr/bin/env python
"""Step time over a long run, per iteration (one HIP event per iteration + the host clock): does the rate of the 50-step bench hold for seconds,
and when it does not, is it the host (one long iteration, the queue drains) or the chip (every step of a period longer)?
python scripts/sustained_probe.py [steps=3000]"""
import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
cg = importlib.import_module("cat-generator_amd")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
cg.manual_seed(1)
CONFIG = int(os.environ.get("CONFIG", "2"))             # BASELINE configs: 2 = G32up-c RGB batch 128, 3 = G32up grayscale batch 256, 5 = 64x64 RGB batch 64
dims, NB = {2: ((3, 32, 32), 128), 3: ((1, 32, 32), 256), 5: ((3, 64, 64), 64)}[CONFIG]
G = cg.models.create_G_decoder_upsampling32(dims, 100) if CONFIG == 3 else cg.models.create_G(dims, 100)
D = cg.models.create_D(dims)
S = cg.adversarial.State(dict(batchSize=NB, seed=1), G, D)
data = cg.adversarial.TrainData(np.random.RandomState(100).rand(512, *dims).astype(np.float32))
for _ in range(10):
    cg.adversarial.iteration(S, data, NB)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
host = np.zeros(steps + 1)
t0 = time.perf_counter()
ev[0].record()
K = int(os.environ.get("IN_FLIGHT", "0"))      # experiment: the host waits for iteration i - K before it enqueues iteration i
for i in range(steps):
    if K and i >= K:
        ev[i + 1 - K].synchronize()
    cg.adversarial.iteration(S, data, NB)
    ev[i + 1].record()
    host[i + 1] = time.perf_counter() - t0
torch.cuda.synchronize()
gpu = np.array([ev[i].elapsed_time(ev[i + 1]) for i in range(steps)])
hst = np.diff(host) * 1e3
blk = 100
print("GPU ms/step per block of 100 :", " ".join(f"{gpu[b:b + blk].mean():.3f}" for b in range(0, steps, blk)))
print("host ms/iter per block of 100:", " ".join(f"{hst[b:b + blk].mean():.2f}" for b in range(0, steps, blk)))
print(f"median GPU step {np.median(gpu):.3f} ms, mean {gpu.mean():.3f}; steps slower than 1.05 x median: {(gpu > 1.05 * np.median(gpu)).sum()} of {steps}")
slow = np.argsort(-hst)[:8]
print("longest host iterations (index, host ms, GPU step ms):", [(int(i), round(float(hst[i]), 1), round(float(gpu[i]), 2)) for i in sorted(slow)])
bad = np.where(gpu > 1.05 * np.median(gpu))[0]
if len(bad):
    print(f"slow GPU steps span iterations {bad.min()}..{bad.max()} ({host[bad.min()]:.1f} s .. {host[bad.max()]:.1f} s after the start); their mean {gpu[bad].mean():.3f} ms, max {gpu[bad].max():.2f} ms")
if len(sys.argv) > 2:      # second argument: profile the host over the iterations [a, b) of a fresh run of the same length, e.g. 1015:1055
    import cProfile, pstats
    a, b = (int(v) for v in sys.argv[2].split(":"))
    for i in range(steps):
        if i == a:
            pr = cProfile.Profile(); pr.enable()
        cg.adversarial.iteration(S, data, NB)
        if i == b - 1:
            pr.disable()
            pstats.Stats(pr).sort_stats("tottime").print_stats(14)
    torch.cuda.synchronize()
