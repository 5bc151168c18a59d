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


# ------------------------------------------------------- lockstep execution of identical branches
class _GroupCtx:
    """Per-branch positions in the counter stream, so that walking several branches layer by layer draws exactly
    the masks a branch-after-branch walk would (and the oracle does)."""

    def __init__(self, cursors):
        self.cur = list(cursors)


def _ptr_array(ptrs):
    import ctypes
    return (ctypes.c_void_p * len(ptrs))(*[int(p) if p else None for p in ptrs])


def _split(Y, G):
    """A stacked tensor ([G*N, ...], branch-major) as its G per-branch slices, each remembering the block it came from."""
    n, N = Y.t.numel() // G, Y.shape[0] // G
    return [Tensor(Y.t[i * n:(i + 1) * n], (N,) + tuple(Y.shape[1:]), Y.fmt, Y.ups, grp=(Y.t, i, G)) for i in range(G)]


def _stacked(xs):
    """The block behind G per-branch tensors if they are exactly its G slices in order (else None)."""
    g0 = xs[0].grp if isinstance(xs[0], Tensor) else None
    if g0 is None or g0[2] != len(xs):
        return None
    x0 = xs[0]
    for i, x in enumerate(xs):
        if (not isinstance(x, Tensor) or x.grp is None or x.grp[0] is not g0[0] or x.grp[1] != i or x.shape != x0.shape
                or x.fmt != x0.fmt or x.ups):
            return None
    return Tensor(g0[0], (len(xs) * x0.shape[0],) + tuple(x0.shape[1:]), x0.fmt)


def _seed_slices(mods, role, shape, fmt):
    """Make the `role` buffers of sibling modules the slices of one block (owned by the first sibling), so that modules
    which must run one launch per branch (own parameters) still hand a stacked tensor to what follows."""
    shape = tuple(int(d) for d in shape)
    n = 1
    for d in shape:
        n *= d
    cur = [m._bufs.get((role, n)) for m in mods]
    if all(c is not None and c.shape == shape and c.fmt == fmt for c in cur) and _stacked(cur) is not None:
        return
    block = mods[0]._get((role, "block"), (len(mods) * shape[0],) + shape[1:], fmt)
    for m, sl in zip(mods, _split(block, len(mods))):
        m._bufs[(role, n)] = sl


def _restack(owner, xs, like):
    """Copy G per-branch tensors into one block (used when a stacked module receives gradients from an unstacked one)."""
    xs = [x if x.fmt == like.fmt else (as_nhwc(x) if like.fmt == "nhwc" else as_plain(x)) for x in xs]
    x0 = xs[0]
    block = owner._get(("restack", "block"), (len(xs) * x0.shape[0],) + tuple(x0.shape[1:]), x0.fmt)
    nb = x0.t.numel() * 4
    for i, x in enumerate(xs):
        lib().memcpy_d2d(stream(), block.ptr + i * nb, x.ptr, nb)
    return block


class _Stackable:
    """Mixin for parameter-free modules: when the inputs of G sibling instances are the slices of one block, the first
    sibling runs ONE launch over the stacked batch and every sibling's output / gradInput is a slice of its result."""
    stacking = True

    @staticmethod
    def _group_forward(mods, inputs, ctx):
        m0 = mods[0]
        X = _stacked(inputs) if _Stackable.stacking else None
        if X is None:
            m0._stk = None
            return Module._group_forward(mods, inputs, ctx)
        ys = _split(m0.updateOutput(X), len(mods))
        m0._stk = ys[0].grp[0]   # the block of this stacked forward; a later per-module forward replaces m0.output
        for m, y in zip(mods, ys):
            m.output = y
        return ys

    @staticmethod
    def _ran_stacked(m0):
        o = m0.output
        return isinstance(o, Tensor) and o.grp is not None and o.grp[0] is getattr(m0, "_stk", None)

    @staticmethod
    def _group_backward(mods, inputs, gouts, scale, acc, ctx):
        m0 = mods[0]
        if not _Stackable._ran_stacked(m0):
            return Module._group_backward(mods, inputs, gouts, scale, acc, ctx)
        X = _stacked(inputs)
        Gd = _stacked(gouts)
        if Gd is None:
            Gd = _restack(m0, gouts, m0.output if isinstance(m0.output, Tensor) else gouts[0])
        gs = _split(m0.updateGradInput(X, Gd), len(mods))
        for m, g in zip(mods, gs):
            m.gradInput = g
        return gs


def group_forward(mods, inputs, ctx):
    return type(mods[0])._group_forward(mods, inputs, ctx)


def group_backward(mods, inputs, gouts, scale, acc, ctx):
    """acc=True: Module:backward (gradInput + accGradParameters); acc=False: updateGradInput only."""
    return type(mods[0])._group_backward(mods, inputs, gouts, scale, acc, ctx)


def _freeze(v):
    if isinstance(v, (list, tuple)):
        return tuple(_freeze(e) for e in v)
    if isinstance(v, dict):
        return tuple(sorted((k, _freeze(e)) for k, e in v.items()))
    return v


def structure_signature(m):
    """Two modules with equal signatures run the same launches with the same geometry."""
    sig = []
    for mod in m.listModules():
        shapes = tuple(getattr(mod, n).shape for n in mod._param_names if getattr(mod, n, None) is not None)
        extra = tuple(getattr(mod, k, None) for k in ("kW", "kH", "padW", "padH", "p", "sizes", "negative_scale", "height",
                                                      "width", "useRotation", "useScale", "useTranslation", "permutations",
                                                      "dimension", "scale_factor", "train"))
        sig.append((type(mod).__name__, shapes, _freeze(extra)))
    return tuple(sig)


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
            b = Tensor(b.t, shape, fmt, grp=b.grp if len(shape) and len(b.shape) and shape[0] == b.shape[0] else None)
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

    # --- lockstep defaults: one call per branch, each at its own position in the counter stream
    @staticmethod
    def _group_forward(mods, inputs, ctx):
        outs = []
        r = rng()
        for b, (m, x) in enumerate(zip(mods, inputs)):
            saved, r.offset = r.offset, ctx.cur[b]
            outs.append(m.updateOutput(x))
            ctx.cur[b], r.offset = r.offset, saved
        return outs

    @staticmethod
    def _group_backward(mods, inputs, gouts, scale, acc, ctx):
        if acc:
            return [m.backward(x, g, scale) for m, x, g in zip(mods, inputs, gouts)]
        return [m.updateGradInput(x, g) for m, x, g in zip(mods, inputs, gouts)]

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
fusion = os.environ.get("CG_FUSION", "1") != "0"
# ^ nn.fusion: nn.Sequential runs chains of modules as fused launches (GEMM epilogues of csrc/gemm.hip, csrc/fused.hip):
#   [conv|linear, PReLU|LeakyReLU]                      -> activation in the GEMM epilogue (both outputs are kept)
#   [PReLU|LeakyReLU, Pool 2x2, (SpatialDropout)]       -> one pass; the two intermediate tensors are never materialised
#   [conv, SpatialBatchNormalization, PReLU] (training) -> batch statistics from the GEMM epilogue, normalise + PReLU in one
#                                                          pass, backward in two passes over (conv output, gradOutput)
#   [View(C*H*W), Linear, (activation)]                 -> the linear layer reads the NHWC map directly (its canonical weight is
#                                                          that of an H x W convolution); no NCHW view is materialised
# Per-element arithmetic is that of the separate modules.  A fused-away intermediate module has .output = None
# (set nn.fusion = False to inspect every module's output).


def _is_gemm(m):
    return isinstance(m, _GemmLayer) and not isinstance(m, SpatialConvolutionUpsample)


def _is_act(m):
    return isinstance(m, (PReLU, LeakyReLU))


def _act_code(m):
    return 1 if isinstance(m, PReLU) else 2


def _act_slope(m):
    return 0.0 if isinstance(m, PReLU) else float(m.negative_scale)


class Sequential(Module):
    def __init__(self):
        super().__init__()
        self.modules = []

    def add(self, m):
        self.modules.append(m)
        self._plan_key = None
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

    # ---- segments: [(kind, first, end)] covering the module list
    def _plan(self):
        mods = self.modules
        key = (fusion, tuple(m.train for m in mods), tuple(id(m) for m in mods))
        if getattr(self, "_plan_key", None) == key:
            return self._plan_v
        plan, i, n = [], 0, len(mods)
        while i < n:
            m, kind, j = mods[i], "one", i + 1
            if fusion:
                nx = mods[i + 1] if i + 1 < n else None
                nx2 = mods[i + 2] if i + 2 < n else None
                if (isinstance(m, SpatialConvolution) and _is_gemm(m) and isinstance(nx, SpatialBatchNormalization) and nx.train
                        and isinstance(nx2, PReLU)):
                    kind, j = "gemm_bn_act", i + 3
                elif _is_gemm(m) and _is_act(nx) and not isinstance(nx2, _Pool2):
                    kind, j = "gemm_act", i + 2
                elif isinstance(m, View) and len(m.sizes) == 1 and type(nx) is Linear and os.environ.get("CG_VIEW_FUSE", "1") != "0":
                    kind, j = ("view_gemm_act", i + 3) if (_is_act(nx2) and not isinstance(mods[i + 3] if i + 3 < n else None, _Pool2)) \
                        else ("view_gemm", i + 2)
                elif _is_act(m) and isinstance(nx, _Pool2):
                    drop = isinstance(nx2, SpatialDropout) and nx2.train and nx2.fixed_noise is None
                    kind, j = "act_pool", i + (3 if drop else 2)
            plan.append((kind, i, j))
            i = j
        self._plan_key, self._plan_v = key, plan
        return plan

    def updateOutput(self, input):
        cur = input
        plan = self._ran = self._plan()
        mods = self.modules
        for kind, i, j in plan:
            if kind == "one":
                cur = mods[i].updateOutput(cur)
            elif kind == "gemm_act":
                cur = _fwd_gemm_act([mods[i]], [mods[i + 1]], [cur], None)[0]
            elif kind == "act_pool":
                cur = _fwd_act_pool([mods[i]], [mods[i + 1]], [mods[i + 2]] if j - i == 3 else None, [cur], None)[0]
            elif kind in ("view_gemm", "view_gemm_act"):
                cur = _fwd_view_gemm([mods[i]], [mods[i + 1]], [mods[i + 2]] if kind == "view_gemm_act" else None, [cur], None)[0]
            else:
                cur = _fwd_gemm_bn_act(mods[i], mods[i + 1], mods[i + 2], cur)
        self.output = cur
        return cur

    def _walk_back(self, input, gradOutput, scale, acc, on_done=None):
        """Segments of the last forward in reverse.  acc: Module:backward (gradInput + accGradParameters), else
        updateGradInput only.  on_done(i): every parameter gradient of modules[i:] is complete."""
        plan = getattr(self, "_ran", None) or [("one", k, k + 1) for k in range(len(self.modules))]
        mods = self.modules
        cur = gradOutput
        for kind, i, j in reversed(plan):
            inp = input if i == 0 else mods[i - 1].output
            if kind == "act_pool" and (getattr(mods[i], "_fused", None) or {}).get("G") == 1:
                cur = _bwd_act_pool([mods[i]], [mods[i + 1]], [mods[i + 2]] if j - i == 3 else None, [cur], scale, acc)[0]
            elif kind == "gemm_bn_act" and getattr(mods[i + 1], "_fused", None) is not None:
                cur = _bwd_gemm_bn_act(mods[i], mods[i + 1], mods[i + 2], inp, cur, scale, acc)
            else:   # "one", "gemm_act" (both outputs exist), or a chain whose forward ran unfused
                for k in range(j - 1, i - 1, -1):
                    mi = inp if k == i else mods[k - 1].output
                    cur = mods[k].backward(mi, cur, scale) if acc else mods[k].updateGradInput(mi, cur)
            if on_done is not None:
                on_done(i)
        self.gradInput = cur
        return cur

    def updateGradInput(self, input, gradOutput):
        return self._walk_back(input, gradOutput, 1.0, False)

    def accGradParameters(self, input, gradOutput, scale=1.0):
        if any(k != "one" for k, _, _ in (getattr(self, "_ran", None) or [])):
            raise NotImplementedError("nn.Sequential:accGradParameters on its own needs the intermediate outputs the fused "
                                      "forward did not keep: call backward(), or set nn.fusion = False")
        cur = gradOutput
        for i in range(len(self.modules) - 1, 0, -1):
            m, prev = self.modules[i], self.modules[i - 1]
            m.accGradParameters(prev.output, cur, scale)
            cur = m.gradInput
        self.modules[0].accGradParameters(input, cur, scale)

    def backward(self, input, gradOutput, scale=1.0):
        return self._walk_back(input, gradOutput, scale, True)

    @staticmethod
    def _group_forward(mods, inputs, ctx):
        cur = list(inputs)
        plan = mods[0]._plan()
        for m in mods:
            m._ran = plan
        for kind, i, j in plan:
            col = lambda k: [m.modules[k] for m in mods]
            if kind == "one":
                cur = group_forward(col(i), cur, ctx)
            elif kind == "gemm_act":
                cur = _fwd_gemm_act(col(i), col(i + 1), cur, ctx)
            elif kind == "act_pool":
                cur = _fwd_act_pool(col(i), col(i + 1), col(i + 2) if j - i == 3 else None, cur, ctx)
            elif kind in ("view_gemm", "view_gemm_act"):
                cur = _fwd_view_gemm(col(i), col(i + 1), col(i + 2) if kind == "view_gemm_act" else None, cur, ctx)
            else:   # not a lockstep case on the path: branch after branch, each at its own stream position
                outs, r = [], rng()
                for b, m in enumerate(mods):
                    saved, r.offset = r.offset, ctx.cur[b]
                    outs.append(_fwd_gemm_bn_act(m.modules[i], m.modules[i + 1], m.modules[i + 2], cur[b]))
                    ctx.cur[b], r.offset = r.offset, saved
                cur = outs
        for m, c in zip(mods, cur):
            m.output = c
        return cur

    @staticmethod
    def _group_backward(mods, inputs, gouts, scale, acc, ctx):
        cur = list(gouts)
        plan = getattr(mods[0], "_ran", None) or [("one", k, k + 1) for k in range(len(mods[0].modules))]
        for kind, i, j in reversed(plan):
            col = lambda k: [m.modules[k] for m in mods]
            inp = list(inputs) if i == 0 else [m.modules[i - 1].output for m in mods]
            fst = [getattr(m.modules[i], "_fused", None) or {} for m in mods] if kind == "act_pool" else []
            if kind == "act_pool" and fst[0].get("G") == len(mods):
                cur = _bwd_act_pool(col(i), col(i + 1), col(i + 2) if j - i == 3 else None, cur, scale, acc)
            elif kind == "act_pool" and all(f.get("G") == 1 for f in fst):   # the forward ran branch after branch
                cur = [_bwd_act_pool([m.modules[i]], [m.modules[i + 1]], [m.modules[i + 2]] if j - i == 3 else None, [g], scale,
                                     acc)[0] for m, g in zip(mods, cur)]
            elif kind == "gemm_bn_act" and getattr(mods[0].modules[i + 1], "_fused", None) is not None:
                cur = [_bwd_gemm_bn_act(m.modules[i], m.modules[i + 1], m.modules[i + 2], x, g, scale, acc)
                       for m, x, g in zip(mods, inp, cur)]
            else:
                for k in range(j - 1, i - 1, -1):
                    mi = inp if k == i else [m.modules[k - 1].output for m in mods]
                    cur = group_backward(col(k), mi, cur, scale, acc, ctx)
        for m, c in zip(mods, cur):
            m.gradInput = c
        return cur

    def __repr__(self):
        return "nn.Sequential {\n  " + "\n  ".join(repr(m).replace("\n", "\n  ") for m in self.modules) + "\n}"


class ConcatTable(Sequential):
    """Every branch sees the input; output is the table of branch outputs; backward sums gradInputs."""

    def updateOutput(self, input):
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

    @staticmethod
    def _group_forward(mods, inputs, ctx):
        per_child = [group_forward([m.modules[j] for m in mods], list(inputs), ctx) for j in range(len(mods[0].modules))]
        outs = [[per_child[j][b] for j in range(len(per_child))] for b in range(len(mods))]
        for m, o in zip(mods, outs):
            m.output = o
        return outs

    @staticmethod
    def _group_backward(mods, inputs, gouts, scale, acc, ctx):
        per_child = [group_backward([m.modules[j] for m in mods], list(inputs), [g[j] for g in gouts], scale, acc, ctx)
                     for j in range(len(mods[0].modules))]
        blocks = [_stacked(c) if (_Stackable.stacking and all(isinstance(t, Tensor) for t in c)) else None for c in per_child]
        if all(b is not None for b in blocks):   # one add per pair of children over the stacked batch
            tot = mods[0]._sum(blocks)
            outs = _split(tot, len(mods)) if tot.grp is None else list(per_child[0])
            for m, o in zip(mods, outs):
                m.gradInput = o
            return outs
        return [m._sum([per_child[j][b] for j in range(len(per_child))]) for b, m in enumerate(mods)]


_CAT_FUSE = os.environ.get("CG_CAT_FUSE", "1") != "0"   # single-launch concat / split / gradient sum


class Concat(Sequential):
    """nn.Concat(2) (models.lua:688-692): branch outputs joined on channels."""

    concurrent = False  # option: branches on separate HIP streams (measured: no gain under graph replay, slower eager)

    def __init__(self, dimension):
        super().__init__()
        assert dimension == 2, "only channel concatenation is on the path"
        self.dimension = dimension
        self._streams = None

    def _fork_join(self, fns):
        """Run the thunks concurrently, one side stream per branch; results in order.  The branches of D32_st3 are
        dozens of tiny kernels each (localisation nets), so their fixed per-kernel costs overlap."""
        if not (self.concurrent and has_gpu() and len(fns) > 1):
            return [f() for f in fns]
        if self._streams is None:
            self._streams = [(torch.cuda.Stream(), torch.cuda.Event()) for _ in fns]
            self._fork_ev = torch.cuda.Event()
        main = torch.cuda.current_stream()
        self._fork_ev.record(main)
        out = []
        for (s, ev), f in zip(self._streams, fns):
            s.wait_event(self._fork_ev)
            with torch.cuda.stream(s):
                out.append(f())
                WGRAD_DEFER.flush()
                ev.record(s)
        for s, ev in self._streams:
            main.wait_event(ev)
        return out

    grouped = True  # identical branches run in lockstep with grouped GEMM launches
    overlap_groups = os.environ.get("CG_CONCAT_OVERLAP", "1") != "0"
    # ^ groups of branches (D32_st3: the three transformer branches in lockstep / the two-convolution branch) run on two
    #   HIP streams: the long chain of launch-bound kernels of the first hides under the big GEMMs of the second

    def _run_groups(self, thunks):
        """thunks[0] on the current stream, the others forked onto side streams and joined; results in order."""
        if not (self.overlap_groups and has_gpu() and len(thunks) > 1):
            return [f() for f in thunks]
        if getattr(self, "_gstreams", None) is None or len(self._gstreams) < len(thunks) - 1:
            self._gstreams = [(torch.cuda.Stream(), torch.cuda.Event()) for _ in thunks[1:]]
            self._gfork = torch.cuda.Event()
        main = torch.cuda.current_stream()
        self._gfork.record(main)
        out = [None] * len(thunks)
        for k, ((st_, ev), f) in enumerate(zip(self._gstreams, thunks[1:]), start=1):
            st_.wait_event(self._gfork)
            with torch.cuda.stream(st_):
                out[k] = f()
                WGRAD_DEFER.flush()
                ev.record(st_)
        out[0] = thunks[0]()
        for (st_, ev), _ in zip(self._gstreams, thunks[1:]):
            main.wait_event(ev)
        return out

    def _branch_groups(self):
        if getattr(self, "_groups", None) is None:
            by_sig = {}
            for i, m in enumerate(self.modules):
                by_sig.setdefault(structure_signature(m), []).append(i)
            self._groups = [g[k:k + 4] for g in by_sig.values() for k in range(0, len(g), 4)]
            self._groups.sort(key=lambda g: g[0])
            self._draws = {}
        return self._groups

    def _forward_branches(self, input):
        """Branch outputs in order.  Groups of identical branches run layer by layer (GEMMs as one grouped launch);
        the first pass at a given input shape runs branch after branch and records how many counter-stream draws each
        branch consumes, so that lockstep passes can place every branch at exactly the same stream position."""
        if not (self.grouped and has_gpu()):
            return self._fork_join([(lambda m=m: as_nhwc(m.updateOutput(input))) for m in self.modules])
        groups = self._branch_groups()
        key = (tuple(input.shape), self.modules[0].train)
        r = rng()
        outs = [None] * len(self.modules)
        if key not in self._draws:
            draws = []
            for i, m in enumerate(self.modules):
                o0 = r.offset
                outs[i] = as_nhwc(m.updateOutput(input))
                draws.append(r.offset - o0)
            self._draws[key] = draws
            return outs
        draws = self._draws[key]
        base = [r.offset + sum(draws[:i]) for i in range(len(self.modules))]
        end = r.offset + sum(draws)
        def run(idxs):
            mods = [self.modules[i] for i in idxs]
            if len(idxs) > 1:
                ctx = _GroupCtx([base[i] for i in idxs])
                res = group_forward(mods, [input] * len(idxs), ctx)
            else:
                r.offset = base[idxs[0]]
                res = [mods[0].updateOutput(input)]
            return [as_nhwc(o) for o in res]

        for idxs, res in zip(groups, self._run_groups([(lambda g=g: run(g)) for g in groups])):
            for i, o in zip(idxs, res):
                outs[i] = o
        r.offset = end
        return outs

    def _backward_branches(self, input, slices, scale, acc):
        if not (self.grouped and has_gpu()):
            if acc:
                return self._fork_join([(lambda m=m, s=s: as_nhwc(m.backward(input, s, scale))) for m, s in slices])
            return self._fork_join([(lambda m=m, s=s: as_nhwc(m.updateGradInput(input, s))) for m, s in slices])
        grads = [None] * len(self.modules)

        def run(idxs):
            mods = [self.modules[i] for i in idxs]
            gs = [slices[i][1] for i in idxs]
            if len(idxs) > 1:
                res = group_backward(mods, [input] * len(idxs), gs, scale, acc, _GroupCtx([0] * len(idxs)))
            else:
                res = [mods[0].backward(input, gs[0], scale) if acc else mods[0].updateGradInput(input, gs[0])]
            return [as_nhwc(g) for g in res]

        groups = self._branch_groups()
        for idxs, res in zip(groups, self._run_groups([(lambda g=g: run(g)) for g in groups])):
            for i, g in zip(idxs, res):
                grads[i] = g
        return grads

    def updateOutput(self, input):
        outs = self._forward_branches(input)
        N, _, H, W = outs[0].shape
        self._sizes = [o.shape[1] for o in outs]
        Ct = sum(self._sizes)
        out = self._get("out", (N, Ct, H, W), "nhwc")
        if fusion and _CAT_FUSE and len(outs) <= 4 and all(c % 4 == 0 for c in self._sizes):   # one launch for all branches
            import ctypes
            lib().concat_channels(stream(), len(outs), _ptr_array([o.ptr for o in outs]), (ctypes.c_int * len(outs))(*self._sizes),
                                  out.ptr, N * H * W)
        else:
            off = 0
            for o, c in zip(outs, self._sizes):
                lib().copy_channels(stream(), o.ptr, out.ptr, N * H * W, c, 0, Ct, off, c)
                off += c
        self.output = out
        return out

    def _slices(self, gradOutput):
        """Channel slices of the gradient, one per branch; the slices of a lockstep group are the parts of one block, so
        that the group's backward can run stacked launches on them."""
        g = as_nhwc(gradOutput)
        N, Ct, H, W = g.shape
        bufs = [None] * len(self.modules)
        if self.grouped and has_gpu() and _Stackable.stacking:
            for idxs in self._branch_groups():
                if len(idxs) > 1 and len({self._sizes[i] for i in idxs}) == 1:
                    block = self._get(("gslice_block", idxs[0]), (len(idxs) * N, self._sizes[idxs[0]], H, W), "nhwc")
                    for i, sl in zip(idxs, _split(block, len(idxs))):
                        bufs[i] = sl
        dst = [bufs[i] if bufs[i] is not None else self._get(("gslice", i), (N, c, H, W), "nhwc") for i, c in enumerate(self._sizes)]
        if fusion and _CAT_FUSE and len(dst) <= 4 and all(c % 4 == 0 for c in self._sizes):
            import ctypes
            lib().split_channels(stream(), len(dst), g.ptr, _ptr_array([d.ptr for d in dst]), (ctypes.c_int * len(dst))(*self._sizes),
                                 N * H * W)
        else:
            off = 0
            for d, c in zip(dst, self._sizes):
                lib().copy_channels(stream(), g.ptr, d.ptr, N * H * W, Ct, off, c, 0, c)
                off += c
        for i, d in enumerate(dst):
            yield self.modules[i], d

    def _accumulate(self, grads):
        gs = [as_nhwc(g) for g in grads]
        if fusion and _CAT_FUSE and 2 <= len(gs) <= 4 and gs[0].phys_numel() % 4 == 0:   # copy + one add per further branch, in one launch
            acc = self._get("gsum", gs[0].shape, "nhwc")
            lib().sum_n(stream(), len(gs), _ptr_array([g.ptr for g in gs]), acc.ptr, acc.phys_numel())
            self.gradInput = acc
            return acc
        acc = None
        for k, g in enumerate(grads):
            g = as_nhwc(g)
            if acc is None:
                acc = self._get("gsum", g.shape, "nhwc")
                acc.copy(g)
            else:
                lib().axpy(stream(), 1.0, g.ptr, acc.ptr, acc.phys_numel())
        self.gradInput = acc
        return acc

    def updateGradInput(self, input, gradOutput):
        return self._accumulate(self._backward_branches(input, list(self._slices(gradOutput)), 1.0, False))

    def accGradParameters(self, input, gradOutput, scale=1.0):
        for m, s in self._slices(gradOutput):
            m.accGradParameters(input, s, scale)

    def backward(self, input, gradOutput, scale=1.0):
        return self._accumulate(self._backward_branches(input, list(self._slices(gradOutput)), scale, True))


# ------------------------------------------------------------- parameterised layers
class _WgradSide:
    """accGradParameters of the convolution / linear layers on a side HIP stream (event fork / join).

    In Module:backward the weight gradient of a layer feeds nothing but the optimiser, while its data gradient is the
    head of the remaining backward chain.  With the weight-gradient GEMMs (and their split reduces) on their own stream the
    memory- and launch-bound kernels of that chain (activation / pooling / batch-norm backward, the localisation nets)
    run beside MFMA-bound work instead of between it.  Opt-in per backward pass: adversarial.iteration() brackets
    MODEL_D:backward / MODEL_G:backward with begin() / join(); a plain module.backward() call stays on one stream.
    MEASURED (MI355X, batch 128, same box A/B): 8.30 ms/step without, 8.46-8.48 ms with - two concurrent GEMM streams cost
    more (L2 / LDS sharing, lower clocks) than the hidden tail kernels save, so it is OFF unless CG_WGRAD_STREAM=1."""
    enabled = os.environ.get("CG_WGRAD_STREAM", "0") != "0"

    def __init__(self):
        self.stream = None
        self.active = False
        self.used = False

    def begin(self):
        self.active = bool(self.enabled and has_gpu())

    def run(self, fn):
        if self.stream is None:
            self.stream, self.ev_fork, self.ev_join = torch.cuda.Stream(), torch.cuda.Event(), torch.cuda.Event()
        cur = torch.cuda.current_stream()
        self.ev_fork.record(cur)               # gradOutput (and the zeroed gradient vector) are complete on `cur`
        self.stream.wait_event(self.ev_fork)
        with torch.cuda.stream(self.stream):   # tensor.stream() / WS follow torch's current stream
            fn()
        self.used = True

    def join(self):
        """Every weight gradient started since begin() is visible to the current stream afterwards."""
        if self.used:
            with torch.cuda.stream(self.stream):
                WGRAD_DEFER.flush()
            self.ev_join.record(self.stream)
            torch.cuda.current_stream().wait_event(self.ev_join)
        self.used = False
        self.active = False


WGRAD_SIDE = _WgradSide()


class _WgradDefer:
    """Deferred split-K reductions of the weight gradients (cg_conv2d_wgrad_grouped_deferred / cg_conv2d_wgrad_flush): between
    begin() and end() a layer's accGradParameters launches only its GEMM (into a workspace the layer owns) and queues the
    reduction; flush() reduces everything queued on the current stream in ONE launch.  ~20 reductions of ~10 us each, every
    one a short dependent chain, then overlap instead of running back to back.  gradWeight / gradBias are complete after
    the flush - adversarial.iteration() brackets MODEL_D:backward / MODEL_G:backward; stream joins flush first."""
    enabled = os.environ.get("CG_WGRAD_DEFER", "1") != "0"

    def __init__(self):
        self.active = False
        self.pending = []

    def begin(self):
        self.active = bool(self.enabled and fusion and has_gpu())

    def workspace(self, owner, nbytes):
        if getattr(owner, "_wg_pending", False):   # the same layer twice before a flush: its partials are still needed
            self.flush_all()
        t = getattr(owner, "_wg_ws", None)
        if t is None or t.numel() < nbytes:
            t = torch.empty(max(int(nbytes), 4096), dtype=torch.uint8, device=device())
            owner._wg_ws = t
        owner._wg_pending = True
        self.pending.append((owner, stream()))
        return t.data_ptr(), t.numel()

    def flush(self):
        """Reduce what is queued on the CURRENT stream."""
        if not self.pending:
            return
        st = stream()
        lib().conv2d_wgrad_flush(st)
        keep = []
        for owner, s_ in self.pending:
            if s_ == st:
                owner._wg_pending = False
            else:
                keep.append((owner, s_))
        self.pending = keep

    def flush_all(self):
        for st in {s_ for _, s_ in self.pending}:   # ABI streams are raw handles: flush each queue on its own stream
            lib().conv2d_wgrad_flush(st)
        for owner, _ in self.pending:
            owner._wg_pending = False
        self.pending = []

    def end(self):
        self.flush()
        if self.pending:
            raise RuntimeError("weight-gradient reductions still queued on a side stream at the end of the backward pass")
        self.active = False


WGRAD_DEFER = _WgradDefer()


class _GemmLayer(Module):
    """Shared by nn.Linear and nn.SpatialConvolution: canonical parameters + packed copies for the kernels."""
    _param_names = ("weight", "bias")

    _map_in = None   # nn.Linear only: (C, H, W) when it consumes an NHWC map directly (a fused nn.View in front of it)

    def _ensure_packed(self):
        ep = self.weight.epoch.v
        mp = self._map_in
        if (getattr(self, "_packed_epoch", None) == ep and getattr(self, "_packed_ptr", None) == self.weight.ptr
                and getattr(self, "_packed_map", None) == mp):
            return
        Cout, Cin, kH, kW = self._wdims()
        n = Cout * Cin * kH * kW
        if getattr(self, "_wf", None) is None or self._wf.numel() != n or (self._wb is None) != (kH * kW == 1):
            self._wf = torch.empty(n, dtype=torch.float32, device=self.weight.t.device)
            self._wb = torch.empty(n, dtype=torch.float32, device=self.weight.t.device) if kH * kW > 1 else None
        pack = lib().pack_conv_weight_map if mp else lib().pack_conv_weight
        pack(stream(), self.weight.ptr, self._wf.data_ptr(), self._wb.data_ptr() if self._wb is not None else None, Cout, Cin, kH, kW)
        self._packed_epoch, self._packed_ptr, self._packed_map = ep, self.weight.ptr, mp

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

    def _epilogue_ok(self, prep):
        """Can this launch take a fused epilogue?  (The skinny <= 4-plane 3x3 kernels and the Winograd path cannot.)"""
        N, Hp, Wp, Cin, Cout, kH, kW, padH, padW, ups = prep[3]
        skinny = ups == 0 and kH == 3 and kW == 3 and padH == 1 and padW == 1 and Cout in (1, 3) and Cin in (64, 128)
        return not skinny

    def _wb_ptr(self):
        # 1x1 / linear: the canonical [out][in] matrix already is the backward operand [K=out][N=in]
        return self._wb.data_ptr() if self._wb is not None else self.weight.ptr

    # Subclasses provide _prep_fwd(input) -> (x, wf_ptr, out, geom), _prep_gin(gradOutput) -> (dy, wb_ptr, gi, geom) or
    # None (not groupable), _prep_acc(gradOutput) -> (x, dy, geom); geom is the 10-tuple the C ABI takes.
    def updateOutput(self, input):
        x, wf, out, a = self._prep_fwd(input)
        ws, wsb = WS.get(lib().conv2d_workspace_bytes(*a))
        lib().conv2d_forward(stream(), x.ptr, wf, self.bias.ptr, out.ptr, *a, ws, wsb)
        self._x, self.output = x, out
        return out

    def accGradParameters(self, input, gradOutput, scale=1.0):
        x, dy, a = self._prep_acc(gradOutput)
        if WGRAD_DEFER.active:
            ws, wsb = WGRAD_DEFER.workspace(self, lib().conv2d_wgrad_workspace_bytes(*a))
            lib().conv2d_wgrad_grouped_deferred(stream(), 1, _ptr_array([x.ptr]), _ptr_array([dy.ptr]), _ptr_array([self.gradWeight.ptr]),
                                                _ptr_array([self.gradBias.ptr]), *a, float(scale), ws, wsb)
            return
        ws, wsb = WS.get(lib().conv2d_wgrad_workspace_bytes(*a))
        lib().conv2d_wgrad(stream(), x.ptr, dy.ptr, self.gradWeight.ptr, self.gradBias.ptr, *a, float(scale), ws, wsb)

    def backward(self, input, gradOutput, scale=1.0):
        if not WGRAD_SIDE.active:
            return super().backward(input, gradOutput, scale)
        g = gradOutput   # layout conversions (if any) happen once, on the current stream, before the fork
        WGRAD_SIDE.run(lambda: self.accGradParameters(input, g, scale))
        return self.updateGradInput(input, g)

    @staticmethod
    def _group_forward(mods, inputs, ctx):
        preps = [m._prep_fwd(x) for m, x in zip(mods, inputs)]
        if len({p[3] for p in preps}) != 1:
            return Module._group_forward(mods, inputs, ctx)
        if _Stackable.stacking and _stacked([p_[2] for p_ in preps]) is None:   # outputs as slices of one block
            _seed_slices(mods, "out", preps[0][2].shape, preps[0][2].fmt)
            preps = [m._prep_fwd(x) for m, x in zip(mods, inputs)]
        a, G = preps[0][3], len(mods)
        ws, wsb = WS.get(lib().conv2d_workspace_bytes_grouped(G, *a))
        lib().conv2d_forward_grouped(stream(), G, _ptr_array([p[0].ptr for p in preps]), _ptr_array([p[1] for p in preps]),
                                     _ptr_array([m.bias.ptr for m in mods]), _ptr_array([p[2].ptr for p in preps]), *a, ws, wsb)
        for m, p_ in zip(mods, preps):
            m._x, m.output = p_[0], p_[2]
        return [p_[2] for p_ in preps]

    @staticmethod
    def _group_backward(mods, inputs, gouts, scale, acc, ctx):
        G = len(mods)
        gin = [m._prep_gin(g) for m, g in zip(mods, gouts)]
        if any(p_ is None for p_ in gin) or len({p_[3] for p_ in gin}) != 1:
            return Module._group_backward(mods, inputs, gouts, scale, acc, ctx)
        if _Stackable.stacking and _stacked([p_[2] for p_ in gin]) is None:
            _seed_slices(mods, "gin", gin[0][2].shape, gin[0][2].fmt)
            gin = [m._prep_gin(g) for m, g in zip(mods, gouts)]
        a = gin[0][3]
        ws, wsb = WS.get(lib().conv2d_workspace_bytes_grouped(G, *a))
        lib().conv2d_forward_grouped(stream(), G, _ptr_array([p_[0].ptr for p_ in gin]), _ptr_array([p_[1] for p_ in gin]),
                                     None, _ptr_array([p_[2].ptr for p_ in gin]), *a, ws, wsb)
        for m, p_ in zip(mods, gin):
            m.gradInput = p_[2]
        if acc:
            accp = [m._prep_acc(g) for m, g in zip(mods, gouts)]
            aa = accp[0][2]

            def wgrad():
                need = lib().conv2d_wgrad_workspace_bytes_grouped(G, *aa)
                ws, wsb = WGRAD_DEFER.workspace(mods[0], need) if WGRAD_DEFER.active else WS.get(need)
                fn = lib().conv2d_wgrad_grouped_deferred if WGRAD_DEFER.active else lib().conv2d_wgrad_grouped
                fn(stream(), G, _ptr_array([p_[0].ptr for p_ in accp]), _ptr_array([p_[1].ptr for p_ in accp]),
                                           _ptr_array([m.gradWeight.ptr for m in mods]), _ptr_array([m.gradBias.ptr for m in mods]),
                                           *aa, float(scale), ws, wsb)
            if WGRAD_SIDE.active:
                WGRAD_SIDE.run(wgrad)
            else:
                wgrad()
        return [p_[2] for p_ in gin]

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
        if self._map_in:   # the canonical [out][C*H*W] matrix is the canonical weight of a C -> out convolution with an H x W kernel
            C, H, W = self._map_in
            return o, C, H, W
        return o, i, 1, 1

    def _fan_in(self):
        return self.weight.shape[1]

    def _prep_fwd(self, input):
        x = to_device(input)
        o = self.weight.shape[0]
        if self._map_in and not (x.dim() == 4 and x.fmt == "nhwc" and not x.ups and tuple(x.shape[1:]) == tuple(self._map_in)):
            self._map_in = None   # used on its own again: the plain [N, in] form
        if self._map_in:
            C, H, W = self._map_in
            N = x.shape[0]
            self._ensure_packed()
            return x, self._wf.data_ptr(), self._get("out", (N, o)), (N, H, W, C, o, H, W, 0, 0, 0)
        x = as_plain(x)
        N, i = x.shape
        self._ensure_packed()
        return x, self._wf.data_ptr(), self._get("out", (N, o)), (N, 1, 1, i, o, 1, 1, 0, 0, 0)

    def _prep_gin(self, gradOutput):
        dy = as_plain(gradOutput)
        N, o = dy.shape
        i = self.weight.shape[1]
        if self._map_in:   # dx comes out NHWC-flattened: wbT[co][(h*W+w)*C + c] (cg_pack_conv_weight_map)
            C, H, W = self._map_in
            return dy, self._wb.data_ptr(), self._get("gin", (N, C, H, W), "nhwc"), (N, 1, 1, o, i, 1, 1, 0, 0, 0)
        return dy, self.weight.ptr, self._get("gin", (N, i)), (N, 1, 1, o, i, 1, 1, 0, 0, 0)

    def _prep_acc(self, gradOutput):
        x, dy = self._x, as_plain(gradOutput)
        if self._map_in:
            C, H, W = self._map_in
            return x, dy, (x.shape[0], H, W, C, self.weight.shape[0], H, W, 0, 0, 0)
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
            return None  # folded-upsampling data gradient: its own entry point, not grouped
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
    _group_forward = staticmethod(Module._group_forward)
    _group_backward = staticmethod(Module._group_backward)

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

    @staticmethod
    def _group_forward(mods, inputs, ctx):
        if _Stackable.stacking and all(isinstance(x, Tensor) and not x.ups for x in inputs):
            _seed_slices(mods, "out", inputs[0].shape, inputs[0].fmt)
        return Module._group_forward(mods, inputs, ctx)

    @staticmethod
    def _group_backward(mods, inputs, gouts, scale, acc, ctx):
        if not _Stackable.stacking:
            return Module._group_backward(mods, inputs, gouts, scale, acc, ctx)
        x0 = mods[0]._x
        X = _stacked([m._x for m in mods])
        n = x0.phys_numel()
        if fusion and X is not None and n % 4 == 0 and len(mods) <= 4:
            # one launch for the G modules: stacked x / dy / dx, one slope and one gradient accumulator per group
            gs = [g if g.fmt == x0.fmt else (as_nhwc(g) if x0.fmt == "nhwc" else as_plain(g)) for g in gouts]
            Gd = _stacked(gs)
            if Gd is None:
                Gd = _restack(mods[0], gs, x0)
            G = len(mods)
            dx = mods[0]._get(("gin", "gblock"), (G * x0.shape[0],) + tuple(x0.shape[1:]), x0.fmt)
            ws, wsb = WS.get(lib().prelu_backward_grouped_workspace_bytes(G, n)) if acc else (None, 0)
            lib().prelu_backward_grouped(stream(), X.ptr, Gd.ptr, _ptr_array([m.weight.ptr for m in mods]), dx.ptr,
                                         _ptr_array([m.gradWeight.ptr for m in mods]) if acc else None, float(scale), G, n,
                                         ws, wsb)
            outs = _split(dx, G)
            for m, o in zip(mods, outs):
                m.gradInput = o
            return outs
        _seed_slices(mods, "gin", x0.shape, x0.fmt)
        return Module._group_backward(mods, inputs, gouts, scale, acc, ctx)


class _Elementwise(Module):
    def _match(self, g, x):
        return g if g.fmt == x.fmt else (as_nhwc(g) if x.fmt == "nhwc" else as_plain(g))


class LeakyReLU(_Stackable, _Elementwise):
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
class View(_Stackable, Module):
    """nn.View(...): the logical NCHW reinterpretation; with NHWC storage this is where the permutation lives."""
    _typename = "nn.View"

    def __init__(self, *sizes):
        super().__init__()
        self.sizes = tuple(int(s) for s in sizes)

    def updateOutput(self, input):
        self._skip = False
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
        if getattr(self, "_skip", False):   # fused into the nn.Linear behind it, whose gradInput already is the NHWC map
            self.gradInput = gradOutput
            return gradOutput
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
            return Tensor(x.t, (N, H, W, C), "plain", 0, x.epoch, grp=x.grp)
        if order == (0, 3, 1, 2):  # BHWD -> NCHW
            assert x.fmt == "plain"
            N, H, W, C = x.shape
            return Tensor(x.t, (N, C, H, W), "nhwc", 0, x.epoch, grp=x.grp)
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


class _Pool2(_Stackable, Module):
    def __init__(self, kW, kH, dW=None, dH=None):
        super().__init__()
        dW, dH = dW or kW, dH or kH
        assert (kW, kH, dW, dH) == (2, 2, 2, 2), "the path only pools 2x2 stride 2"

    # Sibling instances fed the SAME tensor (the localisation nets of D32_st3's three transformer branches all start by
    # pooling the trunk's output, models.lua:843) compute the same thing: run one launch and share the result.
    @staticmethod
    def _group_forward(mods, inputs, ctx):
        m0 = mods[0]
        if (fusion and len(mods) > 1 and isinstance(inputs[0], Tensor) and all(x is inputs[0] for x in inputs[1:])
                and os.environ.get("CG_SHARE_POOL", "1") != "0"):
            y = m0.updateOutput(inputs[0])
            for m in mods:
                m._x, m.output = m0._x, y
            m0._stk, m0._shared_in = None, True
            return [y] * len(mods)
        m0._shared_in = False
        return _Stackable._group_forward(mods, inputs, ctx)

    @staticmethod
    def _group_backward(mods, inputs, gouts, scale, acc, ctx):
        m0 = mods[0]
        if getattr(m0, "_shared_in", False) and isinstance(m0, SpatialAveragePooling):
            gs = [as_nhwc(g) for g in gouts]
            Gd = _stacked(gs)
            if Gd is not None:   # the average pool's backward does not look at its input: one launch over the stacked gradients
                G = len(mods)
                N, C, H, W = m0._x.shape
                gi = m0._get(("gin", "shared"), (G * N, C, H, W), "nhwc")
                lib().avgpool2_backward(stream(), Gd.ptr, gi.ptr, G * N, H, W, C)
                outs = _split(gi, G)
                for m, o in zip(mods, outs):
                    m.gradInput = o
                return outs
        if getattr(m0, "_shared_in", False):
            return Module._group_backward(mods, inputs, gouts, scale, acc, ctx)
        return _Stackable._group_backward(mods, inputs, gouts, scale, acc, ctx)

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


class SpatialDropout(_Stackable, Module):
    """nn.SpatialDropout(p) [upstream, era]: train y = x * mask[n,c] (no rescale); evaluate y = (1-p) x."""
    _typename = "nn.SpatialDropout"

    @staticmethod
    def _group_forward(mods, inputs, ctx):
        """Stacked form: every branch draws its own [N,C] mask at its own position of the counter stream (one tiny
        launch each, into one block), then a single mask multiply runs over the stacked batch."""
        m0 = mods[0]
        X = _stacked(inputs) if _Stackable.stacking else None
        if X is None or any(m.fixed_noise is not None for m in mods):
            m0._stk = None
            return Module._group_forward(mods, inputs, ctx)
        if not m0.train:
            return _Stackable._group_forward(mods, inputs, ctx)
        G = len(mods)
        N, C, H, W = inputs[0].shape
        noise = m0._get(("noise", "block"), (G * N, C))
        r = rng()
        for b in range(G):
            lib().rng_bernoulli_dev(stream(), noise.ptr + 4 * b * N * C, N * C, 1.0 - m0.p, 1.0, r.seed, ctx.cur[b], r.base_ptr())
            ctx.cur[b] += N * C
        out = m0._get("out", X.shape, "nhwc")
        lib().mask_mul(stream(), X.ptr, noise.ptr, out.ptr, G * N, H * W, C, 1)
        ys = _split(out, G)
        m0._stk, m0._noise_block = ys[0].grp[0], noise
        for m, y, nz in zip(mods, ys, _split(noise, G)):
            m.output, m.noise = y, nz
        return ys

    @staticmethod
    def _group_backward(mods, inputs, gouts, scale, acc, ctx):
        m0 = mods[0]
        if not (_Stackable._ran_stacked(m0) and m0.train):
            return _Stackable._group_backward(mods, inputs, gouts, scale, acc, ctx)
        own, m0.noise = m0.noise, m0._noise_block   # the stacked mask for the one stacked multiply
        try:
            return _Stackable._group_backward(mods, inputs, gouts, scale, acc, ctx)
        finally:
            m0.noise = own

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
class AffineTransformMatrixGenerator(_Stackable, Module):
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


class AffineGridGeneratorBHWD(_Stackable, Module):
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

    shared = os.environ.get("CG_SAMPLER_SHARED", "1") != "0"

    @staticmethod
    def _group_forward(mods, inputs, ctx):
        """Sibling transformers sampling the SAME image tensor with stacked grids (D32_st3's branches): one launch over the
        G * N samples (cg_bilinear_sampler_forward_shared); else one launch per branch into the slices of one block."""
        m0, G = mods[0], len(mods)
        m0._shared = None
        if _Stackable.stacking:
            (N, Hi, Wi, C), (_, Ho, Wo, _) = inputs[0][0].shape, inputs[0][1].shape
            grids = _stacked([i_[1] for i_ in inputs])
            img = inputs[0][0]
            if (fusion and BilinearSamplerBHWD.shared and grids is not None and img.fmt == "plain" and grids.fmt == "plain"
                    and all(i_[0].ptr == img.ptr and i_[0].shape == img.shape for i_ in inputs)):
                out = m0._get(("out", "block"), (G * N, Ho, Wo, C))
                lib().bilinear_sampler_forward_shared(stream(), G, img.ptr, grids.ptr, out.ptr, N, Hi, Wi, C, Ho, Wo)
                ys = _split(out, G)
                for m, y in zip(mods, ys):
                    m.output = y
                m0._shared = (out, grids)
                return ys
            _seed_slices(mods, "out", (N, Ho, Wo, C), "plain")
        return Module._group_forward(mods, inputs, ctx)

    @staticmethod
    def _group_backward(mods, inputs, gouts, scale, acc, ctx):
        m0, G = mods[0], len(mods)
        sh = getattr(m0, "_shared", None)
        if (sh is not None and isinstance(m0.output, Tensor) and m0.output.grp is not None and m0.output.grp[0] is sh[0].t
                and all(g.fmt == "plain" for g in gouts)):
            img, grids = inputs[0][0], sh[1]
            (N, Hi, Wi, C), (_, Ho, Wo, _) = img.shape, inputs[0][1].shape
            Gd = _stacked(gouts)
            if Gd is None:
                Gd = _restack(m0, gouts, gouts[0])
            gimg = m0._get(("gimg", "block"), (G * N, Hi, Wi, C))
            ggrid = m0._get(("ggrid", "block"), (G * N, Ho, Wo, 2))
            lib().bilinear_sampler_backward_shared(stream(), G, img.ptr, grids.ptr, Gd.ptr, gimg.ptr, ggrid.ptr, N, Hi, Wi, C, Ho, Wo)
            res = []
            for m, gi, gg in zip(mods, _split(gimg, G), _split(ggrid, G)):
                m.gradInput = [gi, gg]
                res.append(m.gradInput)
            return res
        if _Stackable.stacking:
            _seed_slices(mods, "ggrid", inputs[0][1].shape, "plain")
        return Module._group_backward(mods, inputs, gouts, scale, acc, ctx)


def repack(net):
    """Re-pack the kernel-side weight copies of every convolution / linear layer of `net` whose parameters changed, in ONE
    launch (cg_pack_conv_weight_batch) instead of one launch per layer at its next use.  Layers that run behind a folded
    upsampling keep their own (phase-summed / Winograd) packing.  Called by adversarial.iteration() after each net's
    optimiser step; calling it at any other time is harmless."""
    if not fusion or os.environ.get("CG_BATCH_PACK", "1") == "0":
        return
    todo = []
    for m in net.listModules():
        if isinstance(m, _GemmLayer) and getattr(m, "_wf", None) is not None:
            if getattr(m, "_packed_epoch", None) != m.weight.epoch.v or getattr(m, "_packed_ptr", None) != m.weight.ptr:
                todo.append(m)
    if not todo:
        return
    import ctypes
    n = len(todo)
    dims = [m._wdims() for m in todo]
    ints = lambda k: (ctypes.c_int * n)(*[d[k] for d in dims])
    maps = (ctypes.c_int * n)(*[1 if m._map_in else 0 for m in todo])
    lib().pack_conv_weight_batch(stream(), n, _ptr_array([m.weight.ptr for m in todo]), _ptr_array([m._wf.data_ptr() for m in todo]),
                                 _ptr_array([m._wb.data_ptr() if m._wb is not None else None for m in todo]),
                                 ints(0), ints(1), ints(2), ints(3), maps)
    for m in todo:
        m._packed_epoch, m._packed_ptr, m._packed_map = m.weight.epoch.v, m.weight.ptr, m._map_in


# ------------------------------------------------------------------- fused segments of nn.Sequential
def _fwd_gemm_act(convs, acts, xs, ctx):
    """[conv|linear, PReLU|LeakyReLU] x G branches: one (grouped) GEMM launch whose epilogue writes the pre-activation
    (conv.output, what the activation's backward needs) and the activation (act.output)."""
    G = len(convs)
    preps = [c._prep_fwd(x) for c, x in zip(convs, xs)]
    wino = any(isinstance(c, SpatialConvolution) and p_[0].ups and c._use_wino(p_[0]) for c, p_ in zip(convs, preps))
    ok = (not wino and G <= 4 and len({p_[3] for p_ in preps}) == 1 and all(c._epilogue_ok(p_) for c, p_ in zip(convs, preps)))
    if not ok:
        if G == 1:
            return [acts[0].updateOutput(convs[0].updateOutput(xs[0]))]
        return group_forward(acts, group_forward(convs, xs, ctx), ctx)
    shape, fmt = preps[0][2].shape, preps[0][2].fmt
    if G > 1 and _Stackable.stacking:
        if _stacked([p_[2] for p_ in preps]) is None:
            _seed_slices(convs, "out", shape, fmt)
            preps = [c._prep_fwd(x) for c, x in zip(convs, xs)]
        _seed_slices(acts, "out", shape, fmt)
    ys = [a._get("out", shape, fmt) for a in acts]
    a = preps[0][3]
    code = _act_code(acts[0])
    ws, wsb = WS.get(lib().conv2d_workspace_bytes_grouped(G, *a))
    lib().conv2d_forward_ex(stream(), G, _ptr_array([p_[0].ptr for p_ in preps]), _ptr_array([p_[1] for p_ in preps]),
                            _ptr_array([c.bias.ptr for c in convs]), _ptr_array([p_[2].ptr for p_ in preps]), *a,
                            code, _act_slope(acts[0]), _ptr_array([m.weight.ptr for m in acts]) if code == 1 else None,
                            _ptr_array([y.ptr for y in ys]), None, ws, wsb)
    for c, act, p_, y in zip(convs, acts, preps, ys):
        c._x, c.output = p_[0], p_[2]
        act._x, act.output = p_[2], y
    if G > 1 and isinstance(acts[0], _Stackable):   # let the parameter-free activation run its backward as one stacked launch
        Xs, Ys = _stacked([p_[2] for p_ in preps]), _stacked(ys)
        if Xs is not None and Ys is not None:
            acts[0]._x, acts[0]._stk = Xs, ys[0].grp[0]
        else:
            acts[0]._stk = None
    return ys


def _fwd_view_gemm(views, lins, acts, xs, ctx):
    """[View(C*H*W), Linear, (activation)] x G branches (models.lua:696-698, 849-851): the linear layer consumes the NHWC map
    (cg_pack_conv_weight_map); the NCHW flattening the reference materialises never happens, forward or backward."""
    G = len(views)
    x0 = xs[0]
    ok = isinstance(x0, Tensor) and x0.dim() == 4 and x0.fmt == "nhwc" and not x0.ups
    if ok:
        N, C, H, W = x0.shape
        ok = (H * W <= 64 and C % 16 == 0 and views[0].sizes == (C * H * W,) and lins[0].weight.shape[1] == C * H * W
              and all(isinstance(x, Tensor) and x.shape == x0.shape and x.fmt == "nhwc" and not x.ups for x in xs))
    for v, l in zip(views, lins):
        v._skip, l._map_in = ok, ((C, H, W) if ok else None)
    if ok:
        for v, x in zip(views, xs):
            v.output, v._in_shape = None, x.shape
    else:
        xs = [views[0].updateOutput(xs[0])] if G == 1 else group_forward(views, xs, ctx)
    if acts is not None:
        return _fwd_gemm_act(lins, acts, xs, ctx)
    return [lins[0].updateOutput(xs[0])] if G == 1 else group_forward(lins, xs, ctx)


def _fwd_act_pool(acts, pools, drops, xs, ctx):
    """[PReLU|LeakyReLU, Pool 2x2, (SpatialDropout, training)] x G branches in one pass over the stacked input."""
    G = len(acts)
    a0, p0 = acts[0], pools[0]
    d0 = drops[0] if drops else None
    xs = [x if isinstance(x, Tensor) else to_device(x) for x in xs]
    x0 = xs[0]
    X = None
    if x0.dim() == 4 and x0.fmt == "nhwc" and not x0.ups and G <= 4:
        N, C, H, W = x0.shape
        if C % 4 == 0 and H % 2 == 0 and W % 2 == 0:
            X = x0 if G == 1 else (_stacked(xs) if _Stackable.stacking else None)
    if X is None:   # the separate modules
        a0._fused = None
        chain = [acts, pools] + ([drops] if drops else [])
        cur = xs
        for col in chain:
            cur = [col[0].updateOutput(cur[0])] if G == 1 else group_forward(col, cur, ctx)
        return cur
    code = _act_code(a0)
    last = d0 if d0 is not None else p0
    out = last._get(("out", "fused"), (G * N, C, H // 2, W // 2), "nhwc")
    mask = None
    if d0 is not None:
        mask = d0._get(("noise", "block"), (G * N, C))
        r = rng()
        if G == 1:
            lib().rng_bernoulli_dev(stream(), mask.ptr, N * C, 1.0 - d0.p, 1.0, r.seed, r.take(N * C), r.base_ptr())
        else:
            offs = [ctx.cur[b] for b in range(G)] + [0] * (4 - G)
            lib().rng_bernoulli_dev_grouped(stream(), mask.ptr, N * C, G, 1.0 - d0.p, 1.0, r.seed, *offs, r.base_ptr())
            for b in range(G):
                ctx.cur[b] += N * C
    lib().act_pool2_mask_forward(stream(), X.ptr, out.ptr, mask.ptr if mask is not None else None, G, N, H, W, C, code,
                                 _act_slope(a0), _ptr_array([m.weight.ptr for m in acts]) if code == 1 else None,
                                 1 if isinstance(p0, SpatialMaxPooling) else 0)
    outs = [out] if G == 1 else _split(out, G)
    masks = ([mask] if G == 1 else _split(mask, G)) if mask is not None else [None] * G
    for b in range(G):
        acts[b]._x, acts[b].output = xs[b], None
        pools[b]._x, pools[b].output = None, (outs[b] if d0 is None else None)
        if d0 is not None:
            drops[b].noise, drops[b].output = masks[b], outs[b]
    a0._fused = dict(X=X, mask=mask, G=G, dims=(N, C, H, W))
    return outs


def _bwd_act_pool(acts, pools, drops, gouts, scale, acc):
    a0, p0 = acts[0], pools[0]
    st = a0._fused
    X, mask, G = st["X"], st["mask"], st["G"]
    N, C, H, W = st["dims"]
    gs = [as_nhwc(g) for g in gouts]
    Gd = gs[0] if G == 1 else _stacked(gs)
    if Gd is None:
        Gd = _restack(a0, gs, gs[0])
    dx = a0._get(("gin", "fused"), X.shape, "nhwc")
    code = _act_code(a0)
    want = acc and code == 1
    ws, wsb = WS.get(lib().act_pool2_mask_backward_workspace_bytes(G, N, H, W, C)) if want else (None, 0)
    lib().act_pool2_mask_backward(stream(), X.ptr, Gd.ptr, mask.ptr if mask is not None else None, dx.ptr, G, N, H, W, C, code,
                                  _act_slope(a0), _ptr_array([m.weight.ptr for m in acts]) if code == 1 else None,
                                  _ptr_array([m.gradWeight.ptr for m in acts]) if want else None, float(scale),
                                  1 if isinstance(p0, SpatialMaxPooling) else 0, ws, wsb)
    outs = [dx] if G == 1 else _split(dx, G)
    for b in range(G):
        acts[b].gradInput = outs[b]
        pools[b].gradInput = None
        if drops:
            drops[b].gradInput = None
    return outs


def _fwd_gemm_bn_act(conv, bn, act, input):
    """[conv, SpatialBatchNormalization (training), PReLU] (models.lua:206-208, 212-214, 218-220): the GEMM epilogue leaves
    per-tile column sums of the convolution output, cg_bn_stats_finalize folds them into the batch statistics, and one pass
    normalises and activates.  The normalised tensor is not kept: the backward recomputes it from the convolution output."""
    x = as_nhwc(to_device(input), keep_ups=True)
    C = conv.nOutputPlane
    if C % 4 != 0:
        bn._fused = None
        return act.updateOutput(bn.updateOutput(conv.updateOutput(input)))
    rows, part = 0, None
    if conv._use_wino(x):
        N, Hp, Wp, Ho, Wo = conv._geom(x)
        conv._ensure_packed_ups()
        out = conv._get("out", (N, C, Ho, Wo), "nhwc")
        v = conv._get("wino_v", (lib().conv2d_ups2_wino_v_floats(N, Hp, Wp, conv.nInputPlane),))
        rows = int(lib().conv2d_ups2_wino_stats_rows(N, Hp, Wp, conv.nInputPlane, C))
        if rows:
            part = bn._get("stats_part", (rows, 2, C))
        lib().conv2d_ups2_wino_forward_stats(stream(), x.ptr, conv._u_fwd.data_ptr(), conv.bias.ptr, out.ptr, v.ptr, N, Hp, Wp,
                                             conv.nInputPlane, C, part.ptr if rows else None)
        conv._x, conv.output = x, out
    else:
        xx, wf, out, a = conv._prep_fwd(x)
        rows = int(lib().conv2d_stats_rows(*a)) if conv._epilogue_ok((xx, wf, out, a)) else 0
        ws, wsb = WS.get(lib().conv2d_workspace_bytes(*a))
        if rows:
            part = bn._get("stats_part", (rows, 2, C))
            lib().conv2d_forward_ex(stream(), 1, _ptr_array([xx.ptr]), _ptr_array([wf]), _ptr_array([conv.bias.ptr]),
                                    _ptr_array([out.ptr]), *a, 0, 0.0, None, None, part.ptr, ws, wsb)
        else:
            lib().conv2d_forward(stream(), xx.ptr, wf, conv.bias.ptr, out.ptr, *a, ws, wsb)
        conv._x, conv.output = xx, out
    N, _, H, W = out.shape
    M = N * H * W
    if rows:
        lib().bn_stats_finalize(stream(), part.ptr, rows, C, bn._sums.data_ptr())
    else:
        lib().bn_stats(stream(), out.ptr, M, C, bn._sums.data_ptr())
    bn._count = float(M)
    if parallel.sync_bn_active():
        parallel.allreduce_sum_(bn._sums)
        bn._count = float(M) * parallel.world_size()
    y = act._get("out", out.shape, "nhwc")
    lib().bn_act_forward(stream(), out.ptr, y.ptr, bn.weight.ptr, bn.bias.ptr, bn._sums.data_ptr(), bn._count, M, C,
                         float(bn.eps), float(bn.momentum), bn.running_mean.ptr, bn.running_var.ptr, bn.save_mean.ptr,
                         bn.save_std.ptr, act.weight.ptr)
    bn._x, bn.output = out, None
    act._x, act.output = None, y
    bn._fused = dict(M=M, C=C)
    return y


def _bwd_gemm_bn_act(conv, bn, act, input, gradOutput, scale, acc):
    st = bn._fused
    M, C = st["M"], st["C"]
    x = bn._x
    dy = as_nhwc(gradOutput)
    if getattr(bn, "_bsums3", None) is None:
        dev = bn.weight.t.device
        bn._bsums3 = torch.zeros(2 * C + 1, dtype=torch.float64, device=dev)
        bn._bsums3_g = torch.zeros(2 * C + 1, dtype=torch.float64, device=dev)
    lib().bn_act_backward_stats(stream(), x.ptr, dy.ptr, bn.save_mean.ptr, bn.save_std.ptr, bn.weight.ptr, bn.bias.ptr,
                                act.weight.ptr, M, C, bn._bsums3.data_ptr())
    gs = bn._bsums3
    if parallel.sync_bn_active():
        bn._bsums3_g.copy_(bn._bsums3)
        parallel.allreduce_sum_(bn._bsums3_g)
        gs = bn._bsums3_g
    dx = bn._get("gin", x.shape, "nhwc")
    lib().bn_act_backward(stream(), x.ptr, dy.ptr, bn.weight.ptr, bn.bias.ptr, bn.save_mean.ptr, bn.save_std.ptr,
                          act.weight.ptr, gs.data_ptr(), bn._count, bn._bsums3.data_ptr(), M, C, dx.ptr,
                          bn.gradWeight.ptr if acc else None, bn.gradBias.ptr if acc else None,
                          act.gradWeight.ptr if acc else None, float(scale))
    act.gradInput, bn.gradInput = None, dx
    return conv.backward(input, dx, scale) if acc else conv.updateGradInput(input, dx)


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
