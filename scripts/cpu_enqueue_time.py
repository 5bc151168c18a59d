#!/usr/bin/env python
"""How long the host needs to enqueue one eager training step (no device sync inside): must stay below the GPU time per
step or the eager multi-GPU path becomes launch-bound."""
import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
cg = importlib.import_module("cat-generator_amd")
N = 128
cg.manual_seed(1)
G, D = cg.models.create_G((3, 32, 32), 100), cg.models.create_D((3, 32, 32))
S = cg.adversarial.State(dict(batchSize=N), G, D)
data = cg.adversarial.TrainData(np.random.RandomState(0).rand(512, 3, 32, 32).astype(np.float32))
for _ in range(3):
    cg.adversarial.iteration(S, data, N)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    t0 = time.perf_counter()
    cg.adversarial.iteration(S, data, N)
    ts.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
print("host enqueue ms/step:", " ".join(f"{1e3 * t:.2f}" for t in ts))
if "--profile" in sys.argv:
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(3):
        cg.adversarial.iteration(S, data, N)
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(28)
