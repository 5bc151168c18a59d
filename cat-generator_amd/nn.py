"""Torch7 `nn` module layer re-created over the C ABI (include/catgan.h).

Same class names, constructor signatures and nn.Module protocol the reference
uses: forward/backward/updateOutput/updateGradInput/
accGradParameters, .weight/.bias/.gradWeight/.gradBias/.output/.gradInput/
.modules, getParameters(), training()/evaluate(), listModules(), __typename.
So models.lua's G32up-c / G32up / D32_st3 definitions (models.lua:138-160,
196-228, 640-711, 814-906) translate line for line (see models.py), and the
LuaJIT layer in lua/ is the same code in Lua.

Nothing here computes: every method is one or more launches through the ABI.
Feature maps stay NHWC between modules; nn.View / nn.Transpose / nn.Copy are
the points where the logical Torch7 layout is (re)established.

Two ways through a network:
  * the nn.Module protocol, module by module (updateOutput / updateGradInput / accGradParameters: one C call per
    method) - what any module used on its own gets, and the unfused reference of the tests;
  * planned passes: nn.Sequential:forward / :backward / :updateGradInput (what adversarial.lua calls on MODEL_D / MODEL_G)
    hand the WHOLE tree to the library once (planned.py -> cg_net_add ...) and run as one cg_net_forward / cg_net_backward
    call each; segment fusion, lockstep branches, side streams, deferred reductions and weight re-packing are planned
    below the C ABI (csrc/net.hip).  This file holds no fusion logic.
"""
import os

import numpy as np
import torch

from . import parallel
from .tensor import Tensor, WS, device, has_gpu, lib, rng, stream


# ------------------------------------------------------------------ layout helpers
def materialise(x):
    """Resolve a virtual 2x nearest upsampling into memory."""
    if not x.ups:
        return x
    N, C, H, W = x.shape
    out = Tensor.empty(x.shape, "nhwc")
    lib().upsample2x_forward(stream(), x.ptr, out.ptr, N, H >> 1, W >> 1, C)
    return out


def as_nhwc(x, keep_ups=False):
    if x.fmt == "nhwc":
        return x if (keep_ups or not x.ups) else materialise(x)
    assert x.dim() == 4, f"expected a 4-D feature map, got {x.shape}"
    N, C, H, W = x.shape
    out = Tensor.empty(x.shape, "nhwc")
    lib().nchw_to_nhwc(stream(), x.ptr, out.ptr, N, C, H, W)
    return out


def as_plain(x):
    if x.fmt == "plain":
        return x
    x = materialise(x)
    N, C, H, W = x.shape
    out = Tensor.empty(x.shape, "plain")
    lib().nhwc_to_nchw(stream(), x.ptr, out.ptr, N, C, H, W)
    return out


def to_device(x):
    """Accept host arrays where Torch7 would accept a FloatTensor."""
    if isinstance(x, Tensor):
        return x
    return Tensor.from_numpy(np.asarray(x, dtype=np.float32))


class LazyScalar:
    """A device-resident float (criterion output) that is only copied back when read."""

    def __init__(self, t):
        self.t = t

    def __float__(self):
        return float(self.t.item())

    def __repr__(self):
        return f"{float(self):.6f}"


# ---------------------------------------------------------------------- base class
class Module:
    def __init__(self):
        self.output = None
        self.gradInput = None
        self.train = True
        self._bufs = {}

    @property
    def typename(self):
        return getattr(self, "_typename", "nn." + type(self).__name__)

    def _get(self, key, shape, fmt="plain"):
        # one persistent buffer per (role, size): the half-batch (fake generation) and full-batch passes of the
        # same module keep separate storage, nothing is freed while another stream may still read it
        shape = tuple(int(s) for s in shape)
        n = 1
        for d in shape:
            n *= d
        key = (key, n)
        b = self._bufs.get(key)
        if b is None:
            b = Tensor.empty(shape, fmt)
            self._bufs[key] = b
            return b
        if b.shape != shape or b.fmt != fmt:
            b = Tensor(b.t, shape, fmt)
            self._bufs[key] = b
        return b

    # --- protocol
    def updateOutput(self, input):
        raise NotImplementedError

    def updateGradInput(self, input, gradOutput):
        raise NotImplementedError

    def accGradParameters(self, input, gradOutput, scale=1.0):
        pass

    def forward(self, input):
        return self.updateOutput(input)

    def backward(self, input, gradOutput, scale=1.0):
        self.updateGradInput(input, gradOutput)
        self.accGradParameters(input, gradOutput, scale)
        return self.gradInput

    # --- parameters
    _param_names = ()

    def own_parameters(self):
        """[(module, 'weight', 'gradWeight'), ...] in Torch7 order (weight then bias)."""
        grads = {"weight": "gradWeight", "bias": "gradBias"}
        return [(self, n, grads[n]) for n in self._param_names if getattr(self, n, None) is not None]

    def param_refs(self):
        out = []
        for m in self.listModules():
            out += m.own_parameters()
        return out

    def parameters(self):
        refs = self.param_refs()
        return [getattr(m, p) for m, p, _ in refs], [getattr(m, g) for m, _, g in refs]

    def getParameters(self):
        """Module:getParameters() (train.lua:184-185): one contiguous vector per net, depth-first, weight then
        bias; every weight/bias/gradWeight/gradBias becomes a view into it."""
        from .tensor import Epoch, device
        refs = self.param_refs()
        n = sum(getattr(m, p).nElement() for m, p, _ in refs)
        flat = torch.zeros(n, dtype=torch.float32, device=device())
        gflat = torch.zeros(n, dtype=torch.float32, device=device())
        ep, gep = Epoch(), Epoch()
        off = 0
        for m, p, g in refs:
            w = getattr(m, p)
            k = w.nElement()
            flat[off:off + k].copy_(w.t.reshape(-1))
            setattr(m, p, Tensor(flat[off:off + k], w.shape, "plain", 0, ep))
            setattr(m, g, Tensor(gflat[off:off + k], w.shape, "plain", 0, gep))
            off += k
        ep.bump()
        return Tensor(flat, (n,), "plain", 0, ep), Tensor(gflat, (n,), "plain", 0, gep)

    def zeroGradParameters(self):
        for g in self.parameters()[1]:
            g.zero()

    # --- tree
    def listModules(self):
        return [self]

    def training(self):
        for m in self.listModules():
            m.train = True

    def evaluate(self):
        for m in self.listModules():
            m.train = False

    def clearState(self):
        for m in self.listModules():
            m.output = None
            m.gradInput = None
            m._bufs = {}
        return self

    def float(self):
        return self

    def cuda(self):
        return self

    def type(self, *_):
        return self

    def __repr__(self):
        return self.typename


# ---------------------------------------------------------------------- containers
planned = os.environ.get("CG_PLANNED", "1") != "0"
# ^ nn.planned: a container's :forward / :backward / :updateGradInput run as ONE cg_net_forward / cg_net_backward call - the module
#   tree is described to the library once (planned.py) and everything between the modules (fusion, lockstep branches, streams,
#   re-packing) is planned below the C ABI (csrc/net.hip).  False: the plain per-module walk of the nn.Module protocol.


class Sequential(Module):
    def __init__(self):
        super().__init__()
        self.modules = []
        self._pnet = None
        self._planned_last = False

    def add(self, m):
        self.modules.append(m)
        self._pnet = None
        return self

    def get(self, i):
        return self.modules[i - 1]

    def size(self):
        return len(self.modules)

    def listModules(self):
        out = [self]
        for m in self.modules:
            out += m.listModules()
        return out

    # ---- planned passes (csrc/net.hip): the calls adversarial.lua makes on a whole network
    def _planned_net(self):
        from . import planned as P
        key = tuple(id(m) for m in self.listModules())
        if self._pnet is None or self._pnet[0] != key:
            try:
                self._pnet = (key, P.PlannedNet(self))
            except P.Unsupported:
                self._pnet = (key, None)
        elif self._pnet[1] is not None and any(getattr(m, "fixed_noise", None) is not None for m in self._pnet[1].mods):
            return None   # an explicit dropout mask was attached after the plan was built (test hook): per-module walk
        return self._pnet[1]

    def forward(self, input):
        net = self._planned_net() if (planned and has_gpu() and type(self) is Sequential) else None
        if net is None:
            return self.updateOutput(input)
        x = materialise(to_device(input))
        out = net.forward(x)
        self._planned_last, self._planned_x = True, x
        last = self.modules[-1] if self.modules else None
        if isinstance(last, Copy) and "Cuda" not in last.outtype:   # models.lua:704: the net hands a host FloatTensor back
            out = out.numpy()
        self.output = out
        return out

    def forwardPair(self, input, input2):
        """MODEL:forward(input) now and MODEL:forward(input2) beside it (cg_net_forward_pair; planned nets on a GPU only): returns
        the first output; pairJoin() returns the second, which becomes .output and what :backward continues."""
        net = self._planned_net()
        assert net is not None and planned and has_gpu() and type(self) is Sequential, "forwardPair needs the planned executor"
        x, x2 = materialise(to_device(input)), materialise(to_device(input2))
        out = net.forward_pair(x, x2)
        self._planned_last, self._planned_x = True, x2
        return out

    def pairJoin(self):
        self.output = self._pnet[1].pair_join()
        return self.output

    def _planned_backward(self, gradOutput, scale, acc):
        net = self._pnet[1]
        g = materialise(to_device(gradOutput))
        gi = net.backward(self._planned_x, g, acc, scale)
        first = self.modules[0] if self.modules else None
        if first is not None:
            first.gradInput = gi                                   # adversarial.lua:193 reads MODEL_D.modules[1].gradInput
        if isinstance(first, Copy) and "Cuda" not in first.intype:
            gi = gi.numpy()
        self.gradInput = gi
        return gi

    def module_state(self, m, which="output"):
        """After a planned pass: module m's .output / .gradInput / dropout mask ('noise') as the plan holds it (None when the
        module was fused away).  The per-module walk keeps these on the modules themselves."""
        if not self._planned_last:
            return getattr(m, which)
        return self._pnet[1].module_state(m, {"output": 0, "gradInput": 1, "noise": 2}[which])

    # ---- the nn.Module protocol, module by module
    def updateOutput(self, input):
        self._planned_last = False
        cur = input
        for m in self.modules:
            cur = m.updateOutput(cur)
        self.output = cur
        return cur

    def _walk_back(self, input, gradOutput, scale, acc):
        cur = gradOutput
        for k in range(len(self.modules) - 1, -1, -1):
            mi = input if k == 0 else self.modules[k - 1].output
            cur = self.modules[k].backward(mi, cur, scale) if acc else self.modules[k].updateGradInput(mi, cur)
        self.gradInput = cur
        return cur

    def updateGradInput(self, input, gradOutput):
        if self._planned_last:
            return self._planned_backward(gradOutput, 1.0, False)
        return self._walk_back(input, gradOutput, 1.0, False)

    def accGradParameters(self, input, gradOutput, scale=1.0):
        if self._planned_last:
            raise NotImplementedError("accGradParameters on its own after a planned forward: call backward()")
        cur = gradOutput
        for i in range(len(self.modules) - 1, 0, -1):
            m, prev = self.modules[i], self.modules[i - 1]
            m.accGradParameters(prev.output, cur, scale)
            cur = m.gradInput
        self.modules[0].accGradParameters(input, cur, scale)

    def backward(self, input, gradOutput, scale=1.0):
        if self._planned_last:
            return self._planned_backward(gradOutput, scale, True)
        return self._walk_back(input, gradOutput, scale, True)

    def __repr__(self):
        return "nn.Sequential {\n  " + "\n  ".join(repr(m).replace("\n", "\n  ") for m in self.modules) + "\n}"


class ConcatTable(Sequential):
    """Every branch sees the input; output is the table of branch outputs; backward sums gradInputs."""

    def updateOutput(self, input):
        self._planned_last = False
        self.output = [m.updateOutput(input) for m in self.modules]
        return self.output

    def _sum(self, grads):
        acc = None
        for g in grads:
            if g is None:
                continue
            if acc is None:
                acc = g
            else:
                a, b = as_nhwc(acc) if acc.dim() == 4 else acc, as_nhwc(g) if g.dim() == 4 else g
                out = self._get("sum", a.shape, a.fmt)
                lib().add(stream(), a.ptr, b.ptr, out.ptr, a.phys_numel())
                acc = out
        self.gradInput = acc
        return acc

    def updateGradInput(self, input, gradOutput):
        return self._sum([m.updateGradInput(input, g) for m, g in zip(self.modules, gradOutput)])

    def accGradParameters(self, input, gradOutput, scale=1.0):
        for m, g in zip(self.modules, gradOutput):
            m.accGradParameters(input, g, scale)

    def backward(self, input, gradOutput, scale=1.0):
        return self._sum([m.backward(input, g, scale) for m, g in zip(self.modules, gradOutput)])


class Concat(Sequential):
    """nn.Concat(2) (models.lua:688-692): branch outputs joined on channels."""

    def __init__(self, dimension):
        super().__init__()
        assert dimension == 2, "only channel concatenation is on the path"
        self.dimension = dimension

    def updateOutput(self, input):
        self._planned_last = False
        outs = [as_nhwc(m.updateOutput(input)) for m in self.modules]
        N, _, H, W = outs[0].shape
        self._sizes = [o.shape[1] for o in outs]
        Ct = sum(self._sizes)
        out = self._get("out", (N, Ct, H, W), "nhwc")
        off = 0
        for o, c in zip(outs, self._sizes):
            lib().copy_channels(stream(), o.ptr, out.ptr, N * H * W, c, 0, Ct, off, c)
            off += c
        self.output = out
        return out

    def _slices(self, gradOutput):
        """Channel slices of the gradient, one per branch."""
        g = as_nhwc(gradOutput)
        N, Ct, H, W = g.shape
        off = 0
        for i, (m, c) in enumerate(zip(self.modules, self._sizes)):
            d = self._get(("gslice", i), (N, c, H, W), "nhwc")
            lib().copy_channels(stream(), g.ptr, d.ptr, N * H * W, Ct, off, c, 0, c)
            off += c
            yield m, d

    def _accumulate(self, grads):
        acc = None
        for g in grads:
            g = as_nhwc(g)
            if acc is None:
                acc = self._get("gsum", g.shape, "nhwc")
                acc.copy(g)
            else:
                lib().axpy(stream(), 1.0, g.ptr, acc.ptr, acc.phys_numel())
        self.gradInput = acc
        return acc

    def updateGradInput(self, input, gradOutput):
        return self._accumulate([m.updateGradInput(input, s) for m, s in self._slices(gradOutput)])

    def accGradParameters(self, input, gradOutput, scale=1.0):
        for m, s in self._slices(gradOutput):
            m.accGradParameters(input, s, scale)

    def backward(self, input, gradOutput, scale=1.0):
        return self._accumulate([m.backward(input, s, scale) for m, s in self._slices(gradOutput)])


# ------------------------------------------------------------- parameterised layers
class _GemmLayer(Module):
    """Shared by nn.Linear and nn.SpatialConvolution: canonical parameters + packed copies for the kernels."""
    _param_names = ("weight", "bias")

    def _ensure_packed(self):
        ep = self.weight.epoch.v
        if getattr(self, "_packed_epoch", None) == ep and getattr(self, "_packed_ptr", None) == self.weight.ptr:
            return
        Cout, Cin, kH, kW = self._wdims()
        n = Cout * Cin * kH * kW
        if getattr(self, "_wf", None) is None or self._wf.numel() != n or (self._wb is None) != (kH * kW == 1):
            self._wf = torch.empty(n, dtype=torch.float32, device=self.weight.t.device)
            self._wb = torch.empty(n, dtype=torch.float32, device=self.weight.t.device) if kH * kW > 1 else None
        lib().pack_conv_weight(stream(), self.weight.ptr, self._wf.data_ptr(), self._wb.data_ptr() if self._wb is not None else None,
                               Cout, Cin, kH, kW)
        self._packed_epoch, self._packed_ptr = ep, self.weight.ptr

    def _ensure_packed_ups(self):
        ep = self.weight.epoch.v
        if getattr(self, "_ph_epoch", None) == ep and getattr(self, "_ph_ptr", None) == self.weight.ptr:
            return
        Cout, Cin, kH, kW = self._wdims()
        n = lib().pack_conv_weight_ups2_floats(Cout, Cin, kH, (kH - 1) // 2)
        if getattr(self, "_wf_ph", None) is None or self._wf_ph.numel() != n:
            self._wf_ph = torch.empty(n, dtype=torch.float32, device=self.weight.t.device)
            self._wb_ph = torch.empty(n, dtype=torch.float32, device=self.weight.t.device)
        lib().pack_conv_weight_ups2(stream(), self.weight.ptr, self._wf_ph.data_ptr(), self._wb_ph.data_ptr(), Cout, Cin,
                                    kH, (kH - 1) // 2)
        self._ph_epoch, self._ph_ptr = ep, self.weight.ptr
        if getattr(self, "_wino", False):  # Winograd-domain phase kernels (winograd.hip), same refresh rule
            nu = lib().conv2d_ups2_wino_u_floats(Cin, Cout)
            if getattr(self, "_u_fwd", None) is None or self._u_fwd.numel() != nu:
                self._u_fwd = torch.empty(nu, dtype=torch.float32, device=self.weight.t.device)
                self._u_bwd = torch.empty(nu, dtype=torch.float32, device=self.weight.t.device)
            lib().conv2d_ups2_wino_pack(stream(), self._wf_ph.data_ptr(), self._wb_ph.data_ptr(), self._u_fwd.data_ptr(),
                                        self._u_bwd.data_ptr(), Cout, Cin)

    def _wb_ptr(self):
        # 1x1 / linear: the canonical [out][in] matrix already is the backward operand [K=out][N=in]
        return self._wb.data_ptr() if self._wb is not None else self.weight.ptr

    # Subclasses provide _prep_fwd(input) -> (x, wf_ptr, out, geom), _prep_gin(gradOutput) -> (dy, wb_ptr, gi, geom),
    # _prep_acc(gradOutput) -> (x, dy, geom); geom is the 10-tuple the C ABI takes.
    def updateOutput(self, input):
        x, wf, out, a = self._prep_fwd(input)
        ws, wsb = WS.get(lib().conv2d_workspace_bytes(*a))
        lib().conv2d_forward(stream(), x.ptr, wf, self.bias.ptr, out.ptr, *a, ws, wsb)
        self._x, self.output = x, out
        return out

    def accGradParameters(self, input, gradOutput, scale=1.0):
        x, dy, a = self._prep_acc(gradOutput)
        ws, wsb = WS.get(lib().conv2d_wgrad_workspace_bytes(*a))
        lib().conv2d_wgrad(stream(), x.ptr, dy.ptr, self.gradWeight.ptr, self.gradBias.ptr, *a, float(scale), ws, wsb)

    def reset(self, stdv=None):
        """nn.Linear:reset / nn.SpatialConvolution:reset [upstream]: U(+-stdv*sqrt(3)) if stdv given, else
        U(+-1/sqrt(fan_in)), for weight and bias."""
        s = stdv * np.sqrt(3.0) if stdv is not None else 1.0 / np.sqrt(self._fan_in())
        self.weight.uniform(-s, s)
        self.bias.uniform(-s, s)
        return self


class Linear(_GemmLayer):
    _typename = "nn.Linear"

    def __init__(self, inputSize, outputSize):
        super().__init__()
        self.weight = Tensor.empty((outputSize, inputSize))
        self.bias = Tensor.empty((outputSize,))
        self.gradWeight = Tensor.zeros((outputSize, inputSize))
        self.gradBias = Tensor.zeros((outputSize,))
        self.reset()

    def _wdims(self):
        o, i = self.weight.shape
        return o, i, 1, 1

    def _fan_in(self):
        return self.weight.shape[1]

    def _prep_fwd(self, input):
        x = to_device(input)
        o = self.weight.shape[0]
        x = as_plain(x)
        N, i = x.shape
        self._ensure_packed()
        return x, self._wf.data_ptr(), self._get("out", (N, o)), (N, 1, 1, i, o, 1, 1, 0, 0, 0)

    def _prep_gin(self, gradOutput):
        dy = as_plain(gradOutput)
        N, o = dy.shape
        i = self.weight.shape[1]
        return dy, self.weight.ptr, self._get("gin", (N, i)), (N, 1, 1, o, i, 1, 1, 0, 0, 0)

    def _prep_acc(self, gradOutput):
        x, dy = self._x, as_plain(gradOutput)
        N, i = x.shape
        return x, dy, (N, 1, 1, i, self.weight.shape[0], 1, 1, 0, 0, 0)

    def updateGradInput(self, input, gradOutput):
        dy, wb, gi, a = self._prep_gin(gradOutput)
        ws, wsb = WS.get(lib().conv2d_workspace_bytes(*a))
        lib().conv2d_forward(stream(), dy.ptr, wb, None, gi.ptr, *a, ws, wsb)
        self.gradInput = gi
        return gi

    def __repr__(self):
        return f"nn.Linear({self.weight.shape[1]} -> {self.weight.shape[0]})"


class SpatialConvolution(_GemmLayer):
    """nn.SpatialConvolution(nIn, nOut, kW, kH, dW, dH, padW, padH) — stride 1 only (all the path uses)."""
    _typename = "nn.SpatialConvolution"

    def __init__(self, nInputPlane, nOutputPlane, kW, kH, dW=1, dH=1, padW=0, padH=None):
        super().__init__()
        assert dW == 1 and dH == 1, "the G/D definitions only use stride 1"
        self.nInputPlane, self.nOutputPlane = int(nInputPlane), int(nOutputPlane)
        self.kW, self.kH, self.dW, self.dH = int(kW), int(kH), 1, 1
        self.padW = int(padW)
        self.padH = int(padH if padH is not None else padW)
        self.weight = Tensor.empty((self.nOutputPlane, self.nInputPlane, self.kH, self.kW))
        self.bias = Tensor.empty((self.nOutputPlane,))
        self.gradWeight = Tensor.zeros(self.weight.shape)
        self.gradBias = Tensor.zeros((self.nOutputPlane,))
        self.reset()

    def _wdims(self):
        return self.nOutputPlane, self.nInputPlane, self.kH, self.kW

    def _fan_in(self):
        return self.kW * self.kH * self.nInputPlane

    def _geom(self, x):
        N, C, H, W = x.shape
        assert C == self.nInputPlane, f"{self}: got {C} input planes"
        return N, H >> x.ups, W >> x.ups, H + 2 * self.padH - self.kH + 1, W + 2 * self.padW - self.kW + 1

    def _can_fold_ups(self):
        return self.kH == self.kW and self.kH % 2 == 1 and self.padH == self.padW == (self.kH - 1) // 2

    winograd = os.environ.get("CG_WINOGRAD", "1") != "0"   # F(2x2,3x3) on the phase convolutions of upsample2 -> conv5x5
    winograd_min_tiles = 2048   # below this many 2x2 tiles (N*Hp*Wp/4) the direct kernels win (one workgroup per 64 tiles)

    def _use_wino(self, x):
        """Winograd path (csrc/winograd.hip) for a lazily upsampled input: 5x5, pad 2, even low-res grid, planes % 128."""
        if not (self.winograd and x.ups and self.kH == self.kW == 5 and self.padH == self.padW == 2):
            return False
        N, Hp, Wp, _, _ = self._geom(x)
        if N * Hp * Wp // 4 < self.winograd_min_tiles:
            return False
        ok = bool(lib().conv2d_ups2_wino_supported(N, Hp, Wp, self.nInputPlane, self.nOutputPlane, 5, 2))
        if ok and not getattr(self, "_wino", False):
            self._wino = True
            self._ph_epoch = None   # repack, this time with the Winograd-domain copies
        return ok

    def updateOutput(self, input):
        x = as_nhwc(to_device(input), keep_ups=True)
        if not self._use_wino(x):
            return super().updateOutput(x)
        N, Hp, Wp, Ho, Wo = self._geom(x)
        self._ensure_packed_ups()
        out = self._get("out", (N, self.nOutputPlane, Ho, Wo), "nhwc")
        v = self._get("wino_v", (lib().conv2d_ups2_wino_v_floats(N, Hp, Wp, self.nInputPlane),))
        lib().conv2d_ups2_wino_forward(stream(), x.ptr, self._u_fwd.data_ptr(), self.bias.ptr, out.ptr, v.ptr, N, Hp, Wp,
                                       self.nInputPlane, self.nOutputPlane)
        self._x, self.output = x, out
        return out

    def _prep_fwd(self, input):
        x = as_nhwc(to_device(input), keep_ups=True)
        if x.ups and not self._can_fold_ups():
            x = materialise(x)
        N, Hp, Wp, Ho, Wo = self._geom(x)
        if x.ups:
            self._ensure_packed_ups()
            wf = self._wf_ph.data_ptr()
        else:
            self._ensure_packed()
            wf = self._wf.data_ptr()
        out = self._get("out", (N, self.nOutputPlane, Ho, Wo), "nhwc")
        return x, wf, out, (N, Hp, Wp, self.nInputPlane, self.nOutputPlane, self.kH, self.kW, self.padH, self.padW, x.ups)

    def accGradParameters(self, input, gradOutput, scale=1.0):
        x = self._x
        if not (x.ups and self._use_wino(x)):
            return super().accGradParameters(input, gradOutput, scale)
        # Winograd-domain weight gradient from the transformed input the forward of this batch left in wino_v
        dy = as_nhwc(gradOutput)
        N, Hp, Wp, _, _ = self._geom(x)
        v = self._get("wino_v", (lib().conv2d_ups2_wino_v_floats(N, Hp, Wp, self.nInputPlane),))
        ws, wsb = WS.get(lib().conv2d_ups2_wino_wgrad_workspace_bytes(N, Hp, Wp, self.nInputPlane, self.nOutputPlane))
        lib().conv2d_ups2_wino_wgrad(stream(), v.ptr, dy.ptr, self.gradWeight.ptr, self.gradBias.ptr, N, Hp, Wp,
                                     self.nInputPlane, self.nOutputPlane, float(scale), ws, wsb)

    def _prep_gin(self, gradOutput):
        x = self._x
        if x.ups:
            return None  # folded-upsampling data gradient: its own entry point
        dy = as_nhwc(gradOutput)
        N, Co, Ho, Wo = dy.shape
        gi = self._get("gin", (N, self.nInputPlane, x.shape[2], x.shape[3]), "nhwc")
        return dy, self._wb_ptr(), gi, (N, Ho, Wo, self.nOutputPlane, self.nInputPlane, self.kH, self.kW,
                                        self.kH - 1 - self.padH, self.kW - 1 - self.padW, 0)

    def _prep_acc(self, gradOutput):
        x, dy = self._x, as_nhwc(gradOutput)
        N, Hp, Wp, Ho, Wo = self._geom(x)
        return x, dy, (N, Hp, Wp, self.nInputPlane, self.nOutputPlane, self.kH, self.kW, self.padH, self.padW, x.ups)

    def updateGradInput(self, input, gradOutput):
        x = self._x
        if x.ups:
            # gradient w.r.t. the low-res tensor behind the virtual upsampling (its 2x2 block sum folded in);
            # handed to nn.SpatialUpSamplingNearest as the dual of its lazy output: shape = logical, ups = 1
            dy = as_nhwc(gradOutput)
            N, Hl, Wl = dy.shape[0], x.shape[2], x.shape[3]
            Hp, Wp = Hl >> 1, Wl >> 1
            lo = self._get("gin_lo", (N, self.nInputPlane, Hp, Wp), "nhwc")
            k, pad = self.kH, self.padH
            if self._use_wino(x):
                vdy = self._get("wino_vdy", (lib().conv2d_ups2_wino_v_floats(N, Hp, Wp, 4 * self.nOutputPlane),))
                npart = lib().conv2d_ups2_wino_dgrad_part_floats(N, Hp, Wp, self.nInputPlane, self.nOutputPlane)
                if npart:    # below one workgroup per CU: K slices over blockIdx.z + fixed-order sum (winograd.hip)
                    part = self._get("wino_dpart", (npart,))
                    lib().conv2d_ups2_wino_dgrad_split(stream(), dy.ptr, self._u_bwd.data_ptr(), lo.ptr, vdy.ptr, part.ptr, N, Hp, Wp,
                                                       self.nInputPlane, self.nOutputPlane)
                else:
                    lib().conv2d_ups2_wino_dgrad(stream(), dy.ptr, self._u_bwd.data_ptr(), lo.ptr, vdy.ptr, N, Hp, Wp,
                                                 self.nInputPlane, self.nOutputPlane)
                self.gradInput = Tensor(lo.t, (N, self.nInputPlane, Hl, Wl), "nhwc", 1)
                return self.gradInput
            ws, wsb = WS.get(lib().conv2d_dgrad_ups2_workspace_bytes(N, Hp, Wp, self.nInputPlane, self.nOutputPlane, k, pad))
            lib().conv2d_dgrad_ups2(stream(), dy.ptr, self._wb_ph.data_ptr(), lo.ptr, N, Hp, Wp, self.nInputPlane,
                                    self.nOutputPlane, k, pad, ws, wsb)
            self.gradInput = Tensor(lo.t, (N, self.nInputPlane, Hl, Wl), "nhwc", 1)
            return self.gradInput
        dy, wb, gi, a = self._prep_gin(gradOutput)
        ws, wsb = WS.get(lib().conv2d_workspace_bytes(*a))
        lib().conv2d_forward(stream(), dy.ptr, wb, None, gi.ptr, *a, ws, wsb)
        self.gradInput = gi
        return gi

    def __repr__(self):
        return (f"{self.typename}({self.nInputPlane} -> {self.nOutputPlane}, {self.kW}x{self.kH}, 1,1, "
                f"{self.padW},{self.padH})")


class SpatialConvolutionUpsample(SpatialConvolution):
    """layers/SpatialConvolutionUpsample.lua:1-56: conv to nOut*f^2 planes, then the NCHW buffer
    [N, nOut*f^2, h, w] is *reinterpreted* (a plain view, not a pixel shuffle) as [N, nOut, h*f, w*f]."""
    _typename = "nn.SpatialConvolutionUpsample"
    def __init__(self, nInputPlane, nOutputPlane, kW, kH, factor=2):
        assert kW and kH and nInputPlane and nOutputPlane
        assert kW % 2 == 1, "kW has to be odd"
        assert kH % 2 == 1, "kH has to be odd"
        self.factor = factor
        self.nInputPlaneU, self.nOutputPlaneU = nInputPlane, nOutputPlane
        super().__init__(nInputPlane, nOutputPlane * factor * factor, kW, kH, 1, 1, (kW - 1) // 2)

    def updateOutput(self, input):
        y = as_plain(super().updateOutput(input))  # NCHW memory
        N, _, h, w = y.shape
        self.h, self.w = h, w
        self.output = y.view(N, self.nOutputPlaneU, h * self.factor, w * self.factor)
        return self.output

    def _view_back(self, gradOutput):
        g = as_plain(gradOutput)
        return g.view(g.shape[0], self.nOutputPlaneU * self.factor * self.factor, self.h, self.w)

    def updateGradInput(self, input, gradOutput):
        return super().updateGradInput(input, self._view_back(gradOutput))

    def accGradParameters(self, input, gradOutput, scale=1.0):
        super().accGradParameters(input, self._view_back(gradOutput), scale)


class PReLU(Module):
    """nn.PReLU(nil, nil, true) (models.lua:201): the extra arguments are ignored upstream; one shared slope."""
    _typename = "nn.PReLU"
    _param_names = ("weight",)

    def __init__(self, *_):
        super().__init__()
        self.weight = Tensor.empty((1,)).fill(0.25)
        self.gradWeight = Tensor.zeros((1,))

    def updateOutput(self, input):
        x = materialise(to_device(input))
        out = self._get("out", x.shape, x.fmt)
        lib().prelu_forward(stream(), x.ptr, self.weight.ptr, out.ptr, x.phys_numel())
        self._x = x
        self.output = out
        return out

    def _bwd(self, gradOutput, galpha, scale):
        x = self._x
        dy = gradOutput if gradOutput.fmt == x.fmt else (as_nhwc(gradOutput) if x.fmt == "nhwc" else as_plain(gradOutput))
        gi = self._get("gin", x.shape, x.fmt)
        n = x.phys_numel()
        ws, wsb = WS.get(lib().prelu_backward_workspace_bytes(n)) if galpha else (None, 0)
        lib().prelu_backward(stream(), x.ptr, dy.ptr, self.weight.ptr, gi.ptr, galpha, float(scale), n, ws, wsb)
        self.gradInput = gi
        return gi

    def updateGradInput(self, input, gradOutput):
        return self._bwd(gradOutput, None, 0.0)

    def accGradParameters(self, input, gradOutput, scale=1.0):
        self._bwd(gradOutput, self.gradWeight.ptr, scale)

    def backward(self, input, gradOutput, scale=1.0):  # one fused pass for dx and dalpha
        return self._bwd(gradOutput, self.gradWeight.ptr, scale)

class _Elementwise(Module):
    def _match(self, g, x):
        return g if g.fmt == x.fmt else (as_nhwc(g) if x.fmt == "nhwc" else as_plain(g))


class LeakyReLU(_Elementwise):
    """LeakyReLU.lua:5-31 (negative_scale 0.333; x == 0 takes the positive branch)."""
    _typename = "nn.LeakyReLU"

    def __init__(self, negative_scale=None):
        super().__init__()
        self.negative_scale = negative_scale or 0.333

    def updateOutput(self, input):
        x = materialise(to_device(input))
        out = self._get("out", x.shape, x.fmt)
        lib().leakyrelu_forward(stream(), x.ptr, out.ptr, float(self.negative_scale), x.phys_numel())
        self._x, self.output = x, out
        return out

    def updateGradInput(self, input, gradOutput):
        x = self._x
        dy = self._match(gradOutput, x)
        gi = self._get("gin", x.shape, x.fmt)
        lib().leakyrelu_backward(stream(), x.ptr, dy.ptr, gi.ptr, float(self.negative_scale), x.phys_numel())
        self.gradInput = gi
        return gi


class Sigmoid(_Elementwise):
    _typename = "nn.Sigmoid"

    def updateOutput(self, input):
        x = materialise(to_device(input))
        out = self._get("out", x.shape, x.fmt)
        lib().sigmoid_forward(stream(), x.ptr, out.ptr, x.phys_numel())
        self.output = out
        return out

    def updateGradInput(self, input, gradOutput):
        y = self.output
        dy = self._match(gradOutput, y)
        gi = self._get("gin", y.shape, y.fmt)
        lib().sigmoid_backward(stream(), y.ptr, dy.ptr, gi.ptr, y.phys_numel())
        self.gradInput = gi
        return gi


class SpatialBatchNormalization(Module):
    """nn.SpatialBatchNormalization(n) [upstream]: eps 1e-5, momentum 0.1, affine, gamma ~ U(0,1), beta 0."""
    _typename = "nn.SpatialBatchNormalization"
    _param_names = ("weight", "bias")

    def __init__(self, nFeature, eps=1e-5, momentum=0.1):
        super().__init__()
        self.nFeature, self.eps, self.momentum = int(nFeature), eps, momentum
        self.weight = Tensor.empty((nFeature,)).uniform(0.0, 1.0)
        self.bias = Tensor.zeros((nFeature,))
        self.gradWeight = Tensor.zeros((nFeature,))
        self.gradBias = Tensor.zeros((nFeature,))
        self.running_mean = Tensor.zeros((nFeature,))
        self.running_var = Tensor.empty((nFeature,)).fill(1.0)
        dev = self.weight.t.device
        self._sums = torch.zeros(2 * nFeature, dtype=torch.float64, device=dev)
        self._bsums = torch.zeros(2 * nFeature, dtype=torch.float64, device=dev)
        self._bsums_g = torch.zeros(2 * nFeature, dtype=torch.float64, device=dev)
        self.save_mean = Tensor.zeros((nFeature,))
        self.save_std = Tensor.zeros((nFeature,))  # holds 1/sqrt(var+eps), as THNN's save_std does

    def updateOutput(self, input):
        x = as_nhwc(to_device(input))
        N, C, H, W = x.shape
        M = N * H * W
        out = self._get("out", x.shape, "nhwc")
        if not self.train:
            lib().bn_forward_eval(stream(), x.ptr, out.ptr, self.weight.ptr, self.bias.ptr, self.running_mean.ptr,
                                  self.running_var.ptr, M, C, float(self.eps))
        else:
            lib().bn_stats(stream(), x.ptr, M, C, self._sums.data_ptr())
            self._count = float(M)
            if parallel.sync_bn_active():  # sync-BN: one all-reduce of 2C fp64 sums
                parallel.allreduce_sum_(self._sums)
                self._count = float(M) * parallel.world_size()
            lib().bn_forward(stream(), x.ptr, out.ptr, self.weight.ptr, self.bias.ptr, self._sums.data_ptr(),
                             self._count, M, C, float(self.eps), float(self.momentum), self.running_mean.ptr,
                             self.running_var.ptr, self.save_mean.ptr, self.save_std.ptr)
        self._x, self.output = x, out
        return out

    def _bwd(self, gradOutput, acc, scale):
        x = self._x
        dy = as_nhwc(gradOutput)
        N, C, H, W = x.shape
        M = N * H * W
        assert self.train, "BN backward in evaluate() mode is not on the path"
        lib().bn_backward_stats(stream(), x.ptr, dy.ptr, self.save_mean.ptr, self.save_std.ptr, M, C, self._bsums.data_ptr())
        gs = self._bsums
        if parallel.sync_bn_active():
            self._bsums_g.copy_(self._bsums)
            parallel.allreduce_sum_(self._bsums_g)
            gs = self._bsums_g
        gi = self._get("gin", x.shape, "nhwc")
        lib().bn_backward(stream(), x.ptr, dy.ptr, self.weight.ptr, self.save_mean.ptr, self.save_std.ptr,
                          gs.data_ptr(), self._count, self._bsums.data_ptr(), M, C, gi.ptr,
                          self.gradWeight.ptr if acc else None, self.gradBias.ptr if acc else None, float(scale))
        self.gradInput = gi
        return gi

    def updateGradInput(self, input, gradOutput):
        return self._bwd(gradOutput, False, 0.0)

    def accGradParameters(self, input, gradOutput, scale=1.0):
        self._bwd(gradOutput, True, scale)

    def backward(self, input, gradOutput, scale=1.0):
        return self._bwd(gradOutput, True, scale)

    def __repr__(self):
        return f"nn.SpatialBatchNormalization({self.nFeature})"


# ------------------------------------------------------------ shape / data movement
class View(Module):
    """nn.View(...): the logical NCHW reinterpretation; with NHWC storage this is where the permutation lives."""
    _typename = "nn.View"

    def __init__(self, *sizes):
        super().__init__()
        self.sizes = tuple(int(s) for s in sizes)

    def updateOutput(self, input):
        x = as_plain(to_device(input))
        N = x.shape[0]
        self._in_shape = x.shape
        if len(self.sizes) == 3:
            C, H, W = self.sizes
            out = self._get("out", (N, C, H, W), "nhwc")
            lib().nchw_to_nhwc(stream(), x.ptr, out.ptr, N, C, H, W)
        else:
            out = x.view(N, *self.sizes)
        self.output = out
        return out

    def updateGradInput(self, input, gradOutput):
        self.gradInput = as_plain(gradOutput).view(*self._in_shape)
        return self.gradInput

    def __repr__(self):
        return f"nn.View{self.sizes}"


class Copy(Module):
    """nn.Copy(intype, outtype, forceCopy, dontCast) (models.lua:643,704): the host<->device boundary."""
    _typename = "nn.Copy"

    def __init__(self, intype="torch.FloatTensor", outtype="torch.FloatTensor", forceCopy=False, dontCast=False):
        super().__init__()
        self.intype, self.outtype = intype, outtype

    def _to(self, x, typ):
        if "Cuda" in typ:
            return to_device(x)
        return x.numpy() if isinstance(x, Tensor) else np.asarray(x, dtype=np.float32)

    def updateOutput(self, input):
        self.output = self._to(input, self.outtype)
        return self.output

    def updateGradInput(self, input, gradOutput):
        self.gradInput = self._to(gradOutput, self.intype)
        return self.gradInput


class Transpose(Module):
    """nn.Transpose({a,b},...) — only the NCHW<->BHWD pair the spatial transformer uses (models.lua:870,903);
    with NHWC storage both are relabelings of the same memory."""
    _typename = "nn.Transpose"

    def __init__(self, *perms):
        super().__init__()
        self.permutations = [tuple(p) for p in perms]

    def _apply(self, x, perms):
        order = [0, 1, 2, 3]
        for a, b in perms:
            order[a - 1], order[b - 1] = order[b - 1], order[a - 1]
        order = tuple(order)
        if order == (0, 2, 3, 1):  # NCHW -> BHWD
            x = as_nhwc(to_device(x))
            N, C, H, W = x.shape
            return Tensor(x.t, (N, H, W, C), "plain", 0, x.epoch)
        if order == (0, 3, 1, 2):  # BHWD -> NCHW
            assert x.fmt == "plain"
            N, H, W, C = x.shape
            return Tensor(x.t, (N, C, H, W), "nhwc", 0, x.epoch)
        raise NotImplementedError(f"nn.Transpose{self.permutations}: permutation {order} is not on the path")

    def updateOutput(self, input):
        self.output = self._apply(input, self.permutations)
        return self.output

    def updateGradInput(self, input, gradOutput):
        self.gradInput = self._apply(gradOutput, self.permutations[::-1])
        return self.gradInput


class SpatialUpSamplingNearest(Module):
    """nn.SpatialUpSamplingNearest(2): never materialised when a convolution consumes it (ups flag)."""
    _typename = "nn.SpatialUpSamplingNearest"

    def __init__(self, scale):
        super().__init__()
        assert scale == 2, "the generators only use scale 2"
        self.scale_factor = scale

    def updateOutput(self, input):
        x = as_nhwc(to_device(input))
        N, C, H, W = x.shape
        self.output = Tensor(x.t, (N, C, 2 * H, 2 * W), "nhwc", 1, x.epoch)
        return self.output

    def updateGradInput(self, input, gradOutput):
        if gradOutput.fmt == "nhwc" and gradOutput.ups:  # the consumer conv already folded the 2x2 block sum
            N, C, H2, W2 = gradOutput.shape
            self.gradInput = Tensor(gradOutput.t, (N, C, H2 // 2, W2 // 2), "nhwc", 0)
            return self.gradInput
        g = as_nhwc(gradOutput)
        N, C, H2, W2 = g.shape
        gi = self._get("gin", (N, C, H2 // 2, W2 // 2), "nhwc")
        lib().upsample2x_backward(stream(), g.ptr, gi.ptr, N, H2 // 2, W2 // 2, C)
        self.gradInput = gi
        return gi


class _Pool2(Module):
    def __init__(self, kW, kH, dW=None, dH=None):
        super().__init__()
        dW, dH = dW or kW, dH or kH
        assert (kW, kH, dW, dH) == (2, 2, 2, 2), "the path only pools 2x2 stride 2"

    # Sibling instances fed the SAME tensor (the localisation nets of D32_st3's three transformer branches all start by
    # pooling the trunk's output, models.lua:843) compute the same thing: run one launch and share the result.
    def updateOutput(self, input):
        x = as_nhwc(to_device(input))
        N, C, H, W = x.shape
        out = self._get("out", (N, C, H // 2, W // 2), "nhwc")
        self._fwd(stream(), x.ptr, out.ptr, N, H, W, C)
        self._x, self.output = x, out
        return out


class SpatialAveragePooling(_Pool2):
    _typename = "nn.SpatialAveragePooling"

    def _fwd(self, *a):
        lib().avgpool2_forward(*a)

    def updateGradInput(self, input, gradOutput):
        x, g = self._x, as_nhwc(gradOutput)
        N, C, H, W = x.shape
        gi = self._get("gin", x.shape, "nhwc")
        lib().avgpool2_backward(stream(), g.ptr, gi.ptr, N, H, W, C)
        self.gradInput = gi
        return gi


class SpatialMaxPooling(_Pool2):
    _typename = "nn.SpatialMaxPooling"

    def _fwd(self, *a):
        lib().maxpool2_forward(*a)

    def updateGradInput(self, input, gradOutput):
        x, g = self._x, as_nhwc(gradOutput)
        N, C, H, W = x.shape
        gi = self._get("gin", x.shape, "nhwc")
        lib().maxpool2_backward(stream(), x.ptr, g.ptr, gi.ptr, N, H, W, C)
        self.gradInput = gi
        return gi


class SpatialDropout(Module):
    """nn.SpatialDropout(p) [upstream, era]: train y = x * mask[n,c] (no rescale); evaluate y = (1-p) x."""
    _typename = "nn.SpatialDropout"

    def __init__(self, p=0.5):
        super().__init__()
        self.p = p
        self.noise = None
        self.fixed_noise = None  # test hook: explicit [N,C] mask

    def updateOutput(self, input):
        x = as_nhwc(to_device(input))
        N, C, H, W = x.shape
        out = self._get("out", x.shape, "nhwc")
        if self.train:
            if self.fixed_noise is not None:
                self.noise = to_device(self.fixed_noise)
            else:
                self.noise = self._get("noise", (N, C))
                r = rng()
                lib().rng_bernoulli_dev(stream(), self.noise.ptr, N * C, 1.0 - self.p, 1.0, r.seed, r.take(N * C), r.base_ptr())
            lib().mask_mul(stream(), x.ptr, self.noise.ptr, out.ptr, N, H * W, C, 1)
        else:
            out.copy(x).mul(1.0 - self.p)
        self.output = out
        return out

    def updateGradInput(self, input, gradOutput):
        g = as_nhwc(gradOutput)
        N, C, H, W = g.shape
        gi = self._get("gin", g.shape, "nhwc")
        if self.train:
            lib().mask_mul(stream(), g.ptr, self.noise.ptr, gi.ptr, N, H * W, C, 1)
        else:
            gi.copy(g).mul(1.0 - self.p)
        self.gradInput = gi
        return gi


class Dropout(Module):
    """nn.Dropout(p) v2 [upstream]: train y = x * mask / (1-p); evaluate identity."""
    _typename = "nn.Dropout"

    def __init__(self, p=0.5):
        super().__init__()
        self.p = p
        self.noise = None
        self.fixed_noise = None

    def updateOutput(self, input):
        x = materialise(to_device(input))
        if not self.train:
            self.output = x
            return x
        n = x.phys_numel()
        if self.fixed_noise is not None:
            self.noise = to_device(self.fixed_noise)
        else:
            self.noise = self._get("noise", x.shape, x.fmt)
            r = rng()
            lib().rng_bernoulli_dev(stream(), self.noise.ptr, n, 1.0 - self.p, 1.0 / (1.0 - self.p), r.seed, r.take(n),
                                    r.base_ptr())
        out = self._get("out", x.shape, x.fmt)
        lib().mask_mul(stream(), x.ptr, self.noise.ptr, out.ptr, 1, n, 1, 0)
        self.output = out
        return out

    def updateGradInput(self, input, gradOutput):
        if not self.train:
            self.gradInput = gradOutput
            return gradOutput
        g = gradOutput
        gi = self._get("gin", g.shape, g.fmt)
        lib().mask_mul(stream(), g.ptr, self.noise.ptr, gi.ptr, 1, g.phys_numel(), 1, 0)
        self.gradInput = gi
        return gi


# ----------------------------------------------------------- spatial transformer (stn)
class AffineTransformMatrixGenerator(Module):
    _typename = "nn.AffineTransformMatrixGenerator"

    def __init__(self, useRotation, useScale, useTranslation):
        super().__init__()
        self.useRotation, self.useScale, self.useTranslation = bool(useRotation), bool(useScale), bool(useTranslation)

    def updateOutput(self, input):
        p = as_plain(to_device(input))
        N = p.shape[0]
        out = self._get("out", (N, 2, 3))
        lib().affine_matrix_forward(stream(), p.ptr, out.ptr, N, self.useRotation, self.useScale, self.useTranslation)
        self._p, self.output = p, out
        return out

    def updateGradInput(self, input, gradOutput):
        p = self._p
        gi = self._get("gin", p.shape)
        lib().affine_matrix_backward(stream(), p.ptr, gradOutput.ptr, gi.ptr, p.shape[0], self.useRotation,
                                     self.useScale, self.useTranslation)
        self.gradInput = gi
        return gi


class AffineGridGeneratorBHWD(Module):
    _typename = "nn.AffineGridGeneratorBHWD"

    def __init__(self, height, width):
        super().__init__()
        self.height, self.width = int(height), int(width)

    def updateOutput(self, input):
        T = to_device(input)
        N = T.shape[0]
        out = self._get("out", (N, self.height, self.width, 2))
        lib().affine_grid_forward(stream(), T.ptr, out.ptr, N, self.height, self.width)
        self.output = out
        return out

    def updateGradInput(self, input, gradOutput):
        N = gradOutput.shape[0]
        gi = self._get("gin", (N, 2, 3))
        lib().affine_grid_backward(stream(), gradOutput.ptr, gi.ptr, N, self.height, self.width)
        self.gradInput = gi
        return gi


class BilinearSamplerBHWD(Module):
    """input = {images [N,H,W,C], grids [N,h,w,2]}.  The reference pins this module to the CPU even in GPU mode
    (models.lua:889-899); here it is a device kernel, so `type()` stays the no-op the reference patches in."""
    _typename = "nn.BilinearSamplerBHWD"

    def updateOutput(self, input):
        img, grid = input
        assert img.fmt == "plain" and grid.fmt == "plain"
        N, Hi, Wi, C = img.shape
        _, Ho, Wo, _ = grid.shape
        out = self._get("out", (N, Ho, Wo, C))
        lib().bilinear_sampler_forward(stream(), img.ptr, grid.ptr, out.ptr, N, Hi, Wi, C, Ho, Wo)
        self.output = out
        return out

    def updateGradInput(self, input, gradOutput):
        img, grid = input
        N, Hi, Wi, C = img.shape
        _, Ho, Wo, _ = grid.shape
        g = gradOutput
        assert g.fmt == "plain"
        gimg = self._get("gimg", img.shape)
        ggrid = self._get("ggrid", grid.shape)
        lib().bilinear_sampler_backward(stream(), img.ptr, grid.ptr, g.ptr, gimg.ptr, ggrid.ptr, N, Hi, Wi, C, Ho, Wo)
        self.gradInput = [gimg, ggrid]
        return self.gradInput


# -------------------------------------------------------------------------- criterion
class BCECriterion:
    """nn.BCECriterion() (train.lua:181): sizeAverage, eps 1e-12 [upstream].  forward returns a lazily-read
    device scalar so the hot loop never synchronises on it."""

    def __init__(self):
        self.output = None
        self.gradInput = None
        self._loss = None
        self._g = None

    def _prep(self, input, target):
        x = as_plain(to_device(input))
        t = to_device(target)
        assert x.nElement() == t.nElement()
        return x, t

    def forward(self, input, target):
        x, t = self._prep(input, target)
        if self._loss is None:
            self._loss = torch.zeros(1, dtype=torch.float32, device=x.t.device)
        lib().bce_forward(stream(), x.ptr, t.ptr, self._loss.data_ptr(), x.nElement())
        self.output = LazyScalar(self._loss)
        return self.output

    def backward(self, input, target):
        x, t = self._prep(input, target)
        if self._g is None or self._g.t.numel() != x.nElement():
            self._g = Tensor.empty(x.shape)
        lib().bce_backward(stream(), x.ptr, t.ptr, self._g.ptr, x.nElement())
        self.gradInput = Tensor(self._g.t, x.shape)
        return self.gradInput
