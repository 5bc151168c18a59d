#!/usr/bin/env python
"""Debug: one training step vs the oracle, per-parameter-tensor drift."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
cg = importlib.import_module("cat-generator_amd")
import oracle as O
f32 = np.float32
seed, N = 31, 8
cg.manual_seed(seed); rng = O.RNG(seed)
G, D = cg.models.create_G((3, 32, 32), 100), cg.models.create_D((3, 32, 32))
Go, Do = O.create_G32up_c(3, 100, rng), O.create_D32_st3(3, 32, rng)
S = cg.adversarial.State(dict(batchSize=N, fused_update=True), G, D)
S.keep_outputs = bool(int(os.environ.get("KEEP", "0")))
T = O.Trainer(Go, Do)
rs = np.random.RandomState(9)
pool = rs.rand(32, 3, 32, 32).astype(f32)
data = cg.adversarial.TrainData(pool)
idx = rs.randint(0, 32, size=N // 2)
nd = (rs.rand(N // 2, 100) * 2 - 1).astype(f32); ng = (rs.rand(N, 100) * 2 - 1).astype(f32)
cg.adversarial.iteration(S, data, N, real_idx=idx, noise_D=nd, noise_G=ng)
r = T.step(pool[idx], nd, ng)
for name, net, a, b in (("D", D, S.PARAMETERS_D.numpy(), T.pD), ("G", G, S.PARAMETERS_G.numpy(), T.pG)):
    off = 0
    for m, p, g in net.param_refs():
        n = getattr(m, p).nElement()
        d = np.abs(a[off:off + n] - b[off:off + n])
        if (d > 1e-4).any():
            print(name, off, m, p, getattr(m, p).shape, "max %.2e  n>1e-4 %d of %d" % (d.max(), (d > 1e-4).sum(), n),
                  "first idx", np.nonzero(d > 1e-4)[0][:8])
        off += n
