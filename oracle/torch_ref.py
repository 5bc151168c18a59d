"""PyTorch-CPU (eager, autograd) evaluation of the oracle's module trees.  TEST INFRASTRUCTURE, like the rest of oracle/:
only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may import it; the product never does.

Two jobs:
  * a second, independent pin of the oracle at WHOLE-STEP level (tests/test_oracle_vs_torch.py): the same graphs
    (models.lua:196-228 G32up-c, :138-160 G32up, :640-711 D32_st3, :814-906 spatial transformer) evaluated by PyTorch's
    own CPU kernels and autograd, on the oracle's parameters and dropout masks, must give the oracle's outputs and
    gradients.  PyTorch's conv / batch-norm / pooling / grid-sample code descends from THNN but shares nothing with
    oracle/ops.c or oracle.py.
  * the "best available CPU library" line of bench.py's cpu_baseline: one adversarial.lua:51-275
    iteration timed with PyTorch-CPU eager on the same graphs.

The walker maps each oracle module class to the torch functional of the same Torch7 module; parameters are read from the
oracle modules themselves (no second copy of the model definition).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import oracle as O


class Tape:
    """torch leaves for the oracle's parameter arrays (keyed by module identity + name), in getParameters() order."""

    def __init__(self, net):
        self.leaves, self.order = {}, []
        for m in net.modules():
            for name in getattr(m, "params", []):
                t = torch.tensor(np.array(getattr(m, name), copy=True), requires_grad=True)
                self.leaves[(id(m), name)] = t
                self.order.append(t)

    def p(self, m, name):
        return self.leaves[(id(m), name)]

    def flat_grad(self):
        return np.concatenate([(t.grad if t.grad is not None else torch.zeros_like(t)).reshape(-1).numpy() for t in self.order])

    def zero_grad(self):
        for t in self.order:
            t.grad = None

    def refresh(self, net):
        """Re-read the parameter values after the oracle (or an optimiser) changed them."""
        with torch.no_grad():
            for m in net.modules():
                for name in getattr(m, "params", []):
                    self.leaves[(id(m), name)].copy_(torch.from_numpy(np.asarray(getattr(m, name))))


def _affine_matrix(m, p):
    """AffineTransformMatrixGenerator [stn]: T = R*S*Tr acting on (y, x, 1); parameter order [theta][s][tx, ty]."""
    N = p.shape[0]
    k = 0
    th = torch.zeros(N); sc = torch.ones(N); tx = torch.zeros(N); ty = torch.zeros(N)
    if m.rot:
        th = p[:, k]; k += 1
    if m.scale:
        sc = p[:, k]; k += 1
    if m.trans:
        tx, ty = p[:, k], p[:, k + 1]
    c, s = torch.cos(th), torch.sin(th)
    r0 = torch.stack([c * sc, -s * sc, c * sc * tx - s * sc * ty], dim=1)
    r1 = torch.stack([s * sc, c * sc, s * sc * tx + c * sc * ty], dim=1)
    return torch.stack([r0, r1], dim=1)   # [N, 2, 3]


def _transformer(st, x, tape, masks):
    theta = run(st.loc, x, tape, masks)
    T = _affine_matrix(st.atm, theta)
    base = torch.from_numpy(st.agg.base)                       # [H*W, 3] rows (y_i, x_j, 1)
    g = torch.einsum("pk,nrk->npr", base, T).reshape(x.shape[0], st.agg.H, st.agg.W, 2)   # (y, x) in [-1, 1]
    grid = torch.stack([g[..., 1], g[..., 0]], dim=-1)         # grid_sample wants (x, y)
    return F.grid_sample(x, grid, mode="bilinear", padding_mode="zeros", align_corners=True)


def run(m, x, tape, masks=None):
    """Evaluate oracle module `m` on torch tensor x.  masks: {id(dropout module): ndarray} to reuse the oracle's draws
    (else the dropout modules draw from their own oracle RNG stream, advancing it exactly as the oracle would)."""
    if isinstance(m, O.SpatialTransformer):
        return _transformer(m, x, tape, masks)
    if isinstance(m, O.Concat):
        return torch.cat([run(b, x, tape, masks) for b in m.mods], dim=1)
    if isinstance(m, O.Sequential):
        for c in m.mods:
            x = run(c, x, tape, masks)
        return x
    if isinstance(m, O.Linear):
        return F.linear(x, tape.p(m, "weight"), tape.p(m, "bias"))
    if isinstance(m, O.Conv):
        return F.conv2d(x, tape.p(m, "weight"), tape.p(m, "bias"), padding=m.pad)
    if isinstance(m, O.PReLU):
        return F.prelu(x, tape.p(m, "weight"))
    if isinstance(m, O.LeakyReLU):
        return torch.where(x >= 0, x, x * float(m.s))
    if isinstance(m, O.Sigmoid):
        return torch.sigmoid(x)
    if isinstance(m, O.View):
        return x.reshape(x.shape[0], *m.shape)
    if isinstance(m, O.UpSample2):
        return F.interpolate(x, scale_factor=2, mode="nearest")
    if isinstance(m, O.SBN):
        if m.train:
            return F.batch_norm(x, None, None, tape.p(m, "weight"), tape.p(m, "bias"), training=True, eps=m.eps)
        return F.batch_norm(x, torch.from_numpy(m.running_mean), torch.from_numpy(m.running_var), tape.p(m, "weight"),
                            tape.p(m, "bias"), training=False, eps=m.eps)
    if isinstance(m, O.AvgPool2):
        return F.avg_pool2d(x, 2)
    if isinstance(m, O.MaxPool2):
        return F.max_pool2d(x, 2)
    if isinstance(m, O.SpatialDropout):
        if not m.train:
            return x * (1.0 - m.p)
        mk = masks[id(m)] if masks is not None else m.rng.bernoulli((x.shape[0], x.shape[1]), 1 - m.p, 1.0)
        return x * torch.from_numpy(np.asarray(mk, np.float32))[:, :, None, None]
    if isinstance(m, O.Dropout):
        if not m.train:
            return x
        mk = masks[id(m)] if masks is not None else m.rng.bernoulli(tuple(x.shape), 1 - m.p, 1.0 / (1 - m.p))
        return x * torch.from_numpy(np.asarray(mk, np.float32))
    raise NotImplementedError(type(m).__name__)


def oracle_masks(net):
    """The dropout masks the oracle drew in its last forward, keyed for run(..., masks=)."""
    return {id(m): m.mask for m in net.modules() if isinstance(m, (O.SpatialDropout, O.Dropout)) and getattr(m, "mask", None) is not None}


def bce(out, target):
    """nn.BCECriterion, sizeAverage, eps 1e-12 [upstream]."""
    t = target.reshape(out.shape)
    return -(t * torch.log(out + 1e-12) + (1 - t) * torch.log(1 - out + 1e-12)).mean()


class TorchTrainer:
    """One adversarial.lua:51-275 iteration (D_iterations = G_iterations = 1, Adam, train.lua:26-45 defaults) with
    PyTorch-CPU eager doing every forward / backward; penalty, clamp and Torch7-form Adam as in oracle.Trainer."""

    def __init__(self, G, D, D_L2=1e-4, D_clamp=1.0, G_clamp=5.0):
        self.G, self.D = G, D
        self.pG, _ = O.get_parameters(G)
        self.pD, _ = O.get_parameters(D)
        self.tG, self.tD = Tape(G), Tape(D)
        self.stG, self.stD = {}, {}
        self.o = dict(D_L2=D_L2, D_clamp=D_clamp, G_clamp=G_clamp)

    def step(self, real, noise_d, noise_g):
        half = real.shape[0]
        N = 2 * half
        with torch.no_grad():
            fake = run(self.G, torch.from_numpy(noise_d), self.tG)
        inputs = torch.cat([torch.from_numpy(real), fake], dim=0)
        targets = torch.cat([torch.ones(half), torch.zeros(half)])
        self.tD.zero_grad()
        out = run(self.D, inputs, self.tD)
        bce(out, targets).backward()
        gD = self.tD.flat_grad() + np.float32(self.o["D_L2"]) * self.pD
        np.clip(gD, -self.o["D_clamp"], self.o["D_clamp"], out=gD)
        O.adam(self.pD, gD.astype(np.float32), self.stD)
        self.tD.refresh(self.D)
        self.tG.zero_grad(); self.tD.zero_grad()
        samples = run(self.G, torch.from_numpy(noise_g), self.tG)
        outG = run(self.D, samples, self.tD)
        bce(outG, torch.ones(N)).backward()
        gG = self.tG.flat_grad()
        np.clip(gG, -self.o["G_clamp"], self.o["G_clamp"], out=gG)
        O.adam(self.pG, gG.astype(np.float32), self.stG)
        self.tG.refresh(self.G)
        return dict(fake=fake.numpy(), outD=out.detach().numpy(), gD=gD, gG=gG)
