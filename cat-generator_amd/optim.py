"""Torch7 `optim` functions on flat device vectors (adversarial.lua:240-248, 257-265).

adam follows torch/optim's form [upstream]: m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
x -= lr sqrt(1-b2^t)/(1-b1^t) * m / (sqrt(v) + eps)  — eps is added to sqrt(v) without bias-correcting v.
One fused kernel (cg_adam_step) instead of ~8 elementwise passes; optionally it also applies the L1/L2
penalty and the clamp of adversarial.lua:92-98,110-112 in its prologue (`fused=`).
"""
import torch

from .tensor import Tensor, lib, stream


def adam(opfunc, x, config=None, state=None, fused=None):
    config = config if config is not None else {}
    state = state if state is not None else config
    lr = config.get("learningRate", 0.001)
    beta1 = config.get("beta1", 0.9)
    beta2 = config.get("beta2", 0.999)
    epsilon = config.get("epsilon", 1e-8)

    fx, dfdx = opfunc(x)
    if fx is False:  # the accuracy gate's `return false,false` (adversarial.lua:165): skip the update
        return x, [fx]

    if "t" not in state:
        state["t"] = 0
        state["m"] = Tensor(torch.zeros_like(x.t), x.shape)
        state["v"] = Tensor(torch.zeros_like(x.t), x.shape)
    state["t"] += 1
    l1, l2, clamp = (fused.get("l1", 0.0), fused.get("l2", 0.0), fused.get("clamp", 0.0)) if fused else (0.0, 0.0, 0.0)
    wb = 1 if fused and fused.get("write_back", True) else 0
    if config.get("device_step"):
        # hipGraph replay mode: the step count lives in device memory (kernel arguments are frozen under replay)
        if "t_dev" not in state:
            state["t_dev"] = torch.full((1,), state["t"] - 1, dtype=torch.int64, device=x.t.device)
        lib().counter_add(stream(), state["t_dev"].data_ptr(), 1)
        lib().adam_step_dev(stream(), x.ptr, dfdx.ptr, state["m"].ptr, state["v"].ptr, x.nElement(), lr, beta1, beta2,
                            epsilon, state["t_dev"].data_ptr(), l1, l2, clamp, wb)
    else:
        lib().adam_step(stream(), x.ptr, dfdx.ptr, state["m"].ptr, state["v"].ptr, x.nElement(), lr, beta1, beta2,
                        epsilon, state["t"], l1, l2, clamp, wb)
    x.epoch.bump()
    return x, [fx]


def _fused(fused):
    return ((fused.get("l1", 0.0), fused.get("l2", 0.0), fused.get("clamp", 0.0), 1 if fused.get("write_back", True) else 0)
            if fused else (0.0, 0.0, 0.0, 0))


def sgd(opfunc, x, config=None, state=None, fused=None):
    """optim.sgd as train.lua:201-204 configures it (learningRate, momentum; Torch7 defaults: dampening = momentum,
    no Nesterov, no weight decay, no lr decay).  First step with momentum: v = g [upstream: dfdx:clone()]."""
    config = config if config is not None else {}
    state = state if state is not None else config
    lr = config.get("learningRate", 1e-3)
    mom = config.get("momentum", 0.0)
    damp = config.get("dampening", mom)
    fx, dfdx = opfunc(x)
    if fx is False:
        return x, [fx]
    l1, l2, clamp, wb = _fused(fused)
    first = "dfdx" not in state
    if mom != 0 and first:
        state["dfdx"] = Tensor(torch.zeros_like(x.t), x.shape)
    state["evalCounter"] = state.get("evalCounter", 0) + 1
    lib().sgd_step(stream(), x.ptr, dfdx.ptr, state["dfdx"].ptr if mom != 0 else None, x.nElement(), lr, mom,
                   0.0 if (mom != 0 and first) else damp, l1, l2, clamp, wb)
    x.epoch.bump()
    return x, [fx]


def adagrad(opfunc, x, config=None, state=None, fused=None):
    """optim.adagrad (train.lua:193-196): paramVariance += g^2; x -= lr * g / (sqrt(paramVariance) + 1e-10)."""
    config = config if config is not None else {}
    state = state if state is not None else config
    lr = config.get("learningRate", 1e-3)
    fx, dfdx = opfunc(x)
    if fx is False:
        return x, [fx]
    l1, l2, clamp, wb = _fused(fused)
    if "paramVariance" not in state:
        state["paramVariance"] = Tensor(torch.zeros_like(x.t), x.shape)
    state["evalCounter"] = state.get("evalCounter", 0) + 1
    lib().adagrad_step(stream(), x.ptr, dfdx.ptr, state["paramVariance"].ptr, x.nElement(), lr, l1, l2, clamp, wb)
    x.epoch.bump()
    return x, [fx]


class ConfusionMatrix:
    """optim.ConfusionMatrix(CLASSES) for the binary case of adversarial.lua:74,101-106,285-289, kept on the
    device: counts[pred][target] are only copied back when read."""

    def __init__(self, classes):
        from .tensor import device
        self.classes = classes
        self.counts = torch.zeros(4, dtype=torch.int32, device=device())
        self.totalValid = 0.0

    def zero(self):
        self.counts.zero_()
        self._last_global = None

    def batchAdd(self, outputs, targets):
        lib().confusion_update(stream(), outputs.ptr, targets.ptr, self.counts.data_ptr(), targets.nElement())

    def _global_counts(self):
        """counts summed over the data-parallel ranks (every rank sees 1/R of the batch)."""
        from . import parallel
        import numpy as np
        return np.asarray(parallel.allreduce_sum_host(self.counts.cpu().numpy().tolist()))

    def updateValids(self):
        """Collective under data parallelism: every rank calls it (adversarial.train does, at the end of the epoch)."""
        self._last_global = self._global_counts()
        c = self._last_global.astype(float)
        tot = c.sum()
        self.totalValid = float((c[0] + c[3]) / tot) if tot > 0 else 0.0
        return self.totalValid

    def __repr__(self):  # never a collective: the counts of the last updateValids(), else this rank's own
        g = getattr(self, "_last_global", None)
        c = (g if g is not None else self.counts.cpu().numpy()).astype(int)
        return f"ConfusionMatrix(pred0/t0={c[0]}, pred0/t1={c[1]}, pred1/t0={c[2]}, pred1/t1={c[3]})"
