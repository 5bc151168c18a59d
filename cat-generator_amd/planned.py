"""Host side of the planned executor (cg_net_*, include/catgan.h; csrc/net.hip).

An nn container's :forward / :backward / :updateGradInput (the calls adversarial.lua:84-89,182-197 makes on MODEL_D / MODEL_G)
do not walk the module tree here: the tree is DESCRIBED to the library once - one cg_net_add per module, the same
constructor arguments models.lua passes - and every pass is one cg_net_forward / cg_net_backward call.  Planning (fused
segments, lockstep branches, side streams, deferred reductions, weight re-packing, collectives) happens below the C ABI; this
file only marshals: module tree -> builder calls, parameter tensors -> cg_net_bind, tensors in / out.  lua/catgan/net.lua is
the same marshalling in LuaJIT.

The per-module protocol (updateOutput / updateGradInput / accGradParameters on a single module, nn.py) stays available - it is
what the reference's nn.Module API promises for any module used on its own - and doubles as the unfused reference the planned
passes are tested against.
"""
import contextlib
import ctypes
import os
import weakref

import torch

from . import _abi, parallel
from .tensor import Tensor, device, has_gpu, lib, rng, stream

KIND = dict(Sequential=0, Concat=1, ConcatTable=2, Linear=3, SpatialConvolution=4, PReLU=5, LeakyReLU=6, Sigmoid=7,
            SpatialBatchNormalization=8, View=9, Copy=10, Transpose=11, SpatialUpSamplingNearest=12, SpatialAveragePooling=13,
            SpatialMaxPooling=14, SpatialDropout=15, Dropout=16, AffineTransformMatrixGenerator=17, AffineGridGeneratorBHWD=18,
            BilinearSamplerBHWD=19)


LIVE = weakref.WeakSet()   # every PlannedNet alive (tools/abi_record.py names their handles in a recorded trace)


class Unsupported(Exception):
    """The tree holds a module the planned executor has no entry for: the container falls back to the per-module walk."""


def describe(m):
    """(kind, iargs, fargs) of one module - the arguments of its reference constructor."""
    n = type(m).__name__
    if n not in KIND:
        raise Unsupported(n)
    k = KIND[n]
    if n == "Concat":
        return k, [m.dimension], []
    if n == "Linear":
        return k, [m.weight.shape[1], m.weight.shape[0]], []
    if n == "SpatialConvolution":
        return k, [m.nInputPlane, m.nOutputPlane, m.kW, m.kH, m.padW, m.padH, m.dW, m.dH], []
    if n == "LeakyReLU":
        return k, [], [m.negative_scale]
    if n == "SpatialBatchNormalization":
        return k, [m.nFeature], [m.eps, m.momentum]
    if n == "View":
        if len(m.sizes) not in (1, 3):
            raise Unsupported("nn.View with %d sizes" % len(m.sizes))
        return k, list(m.sizes), []
    if n == "Transpose":
        order = [0, 1, 2, 3]
        for a, b in m.permutations:
            order[a - 1], order[b - 1] = order[b - 1], order[a - 1]
        if tuple(order) == (0, 2, 3, 1):
            return k, [0], []
        if tuple(order) == (0, 3, 1, 2):
            return k, [1], []
        raise Unsupported("nn.Transpose%s" % (m.permutations,))
    if n in ("SpatialDropout", "Dropout"):
        if m.fixed_noise is not None:
            raise Unsupported("explicit dropout mask (test hook): per-module walk")
        return k, [], [m.p]
    if n == "AffineTransformMatrixGenerator":
        return k, [int(m.useRotation), int(m.useScale), int(m.useTranslation)], []
    if n == "AffineGridGeneratorBHWD":
        return k, [m.height, m.width], []
    if n == "SpatialUpSamplingNearest":
        return k, [m.scale_factor], []
    return k, [], []


class _Raw:
    """A device address as an array-interface object, so torch can view memory the library allocated."""

    def __init__(self, ptr, n, typestr="<f4"):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 2}


class PlannedNet:
    """One cg_net handle for a module tree."""

    def __init__(self, root, trace=False):
        self.L = lib()
        self.root = root
        self.h = ctypes.c_void_p()
        self.L.net_create(ctypes.byref(self.h))
        self.trace = trace
        self.blocks = []          # device memory of the plan's buffers (torch owns it: it shows up in torch's accounting)
        self.ids = {}             # id(module) -> cg_net module id
        self.mods = []
        self.desc = []            # the cg_net_add calls that built the tree (replayable: tools/abi_record.py)
        LIVE.add(self)
        if trace:
            self.L.net_set_option(self.h, b"trace", 1)
        elif os.environ.get("CG_NET_ALLOC", "torch") == "lib":
            pass                  # buffers from the library's own allocator (what a LuaJIT host gets; recorded steps use it)
        else:
            self._alloc_cb = _abi.ALLOC_FN(self._alloc)
            self.L.net_set_allocator(self.h, ctypes.cast(self._alloc_cb, ctypes.c_void_p), None)
        self._hook_cb = _abi.HOOK_FN(self._hook)
        self.L.net_set_hook(self.h, ctypes.cast(self._hook_cb, ctypes.c_void_p), None)
        self._add(root, -1)
        self._bound = None
        self._train = None
        self._epochs = None
        self._dp = None
        self.pending = []         # gloo / torch.distributed bucket all-reduces in flight (host-hook transport)

    def __del__(self):
        try:
            self.L.net_destroy(self.h)
        except Exception:
            pass

    # ---- building
    def _add(self, m, parent):
        kind, ia, fa = describe(m)
        ia_c = (ctypes.c_long * max(len(ia), 1))(*[int(v) for v in ia])
        fa_c = (ctypes.c_float * max(len(fa), 1))(*[float(v) for v in fa])
        mid = ctypes.c_int(-1)
        self.L.net_add(self.h, parent, kind, ia_c, len(ia), ctypes.cast(fa_c, ctypes.c_void_p), len(fa), ctypes.byref(mid))
        self.ids[id(m)] = mid.value
        self.mods.append(m)
        self.desc.append((parent, kind, [int(v) for v in ia], [float(v) for v in fa]))
        for c in getattr(m, "modules", []):
            self._add(c, mid.value)

    def _alloc(self, _user, nbytes):
        t = torch.zeros(int(nbytes), dtype=torch.uint8, device=device())
        self.blocks.append(t)
        return t.data_ptr()

    def _view(self, ptr, numel, dtype=torch.float32):
        """torch view of `numel` elements at a device address inside one of the plan's blocks (or None)."""
        for b in self.blocks:
            base = b.data_ptr()
            if base <= ptr < base + b.numel():
                off = ptr - base
                return b[off:off + numel * (8 if dtype == torch.float64 else 4)].view(dtype)
        return None

    def _hook(self, _user, what, buf, count, dtype, _stream):
        """Transport of the plan's collectives when no cg_comm_* communicator is bound (gloo tests, CG_COMM=torch).  The plan hands
        over the HIP stream the exchange belongs on (a branch group's side stream, the weight-gradient stream a bucket's gradients
        were produced on): torch's collective is issued with that stream current, so it is ordered against the kernels that produce
        and consume the buffer."""
        try:
            ctx = contextlib.nullcontext()
            if _stream and torch.cuda.is_available() and int(_stream) != int(stream() or 0):
                ctx = torch.cuda.stream(torch.cuda.ExternalStream(int(_stream)))
            with ctx:
                if what == 0:
                    t = self._view(buf, count, torch.float64 if dtype == 1 else torch.float32)
                    if t is None:      # the library's own allocator (CG_NET_ALLOC=lib): wrap the raw device address
                        t = torch.as_tensor(_Raw(buf, count, "<f8" if dtype == 1 else "<f4"), device=device())
                    parallel.allreduce_sum_torch(t)
                else:
                    g = self._grad_view(buf, count)
                    self.pending.append(parallel.allreduce_mean_async_torch(g))
            return 0
        except Exception as e:   # no exception may cross the C ABI
            import traceback
            traceback.print_exc()
            self._hook_error = e
            return 1

    def _grad_view(self, ptr, count):
        """The slice [ptr, ptr + 4 count) of the flat gradient vector the modules' gradWeight / gradBias are views of."""
        for m in self.mods:
            for g in (getattr(m, "gradWeight", None), getattr(m, "gradBias", None)):
                if isinstance(g, Tensor):
                    flat = g.t._base if g.t._base is not None else g.t
                    base = flat.data_ptr()
                    if base <= ptr < base + flat.numel() * 4:
                        off = (ptr - base) // 4
                        return flat.reshape(-1)[off:off + count]
        raise RuntimeError("gradient bucket outside every bound gradient tensor")

    # ---- per-pass synchronisation of what the host may have changed
    def _sync(self):
        L, h = self.L, self.h
        bound, epochs = [], 0
        seen = set()
        for m in self.mods:
            for slot, (p, g) in enumerate((("weight", "gradWeight"), ("bias", "gradBias"))):
                w = getattr(m, p, None)
                if isinstance(w, Tensor) and p in getattr(m, "_param_names", ()):
                    gw = getattr(m, g)
                    bound.append((self.ids[id(m)], slot, w.ptr, gw.ptr))
                    if id(w.epoch) not in seen:
                        seen.add(id(w.epoch))
                        epochs += w.epoch.v
            if type(m).__name__ == "SpatialBatchNormalization":
                bound.append((self.ids[id(m)], 2, m.running_mean.ptr, m.running_var.ptr))
        if bound != self._bound:
            for mid, slot, a, b in bound:
                L.net_bind(h, mid, slot, a, b)
            self._bound = bound
            self._epochs = None
            if self.trace:   # name the host-owned tensors in the trace
                reg = set()
                for m in self.mods:
                    for nme in ("weight", "gradWeight", "bias", "gradBias", "running_mean", "running_var"):
                        t = getattr(m, nme, None)
                        if isinstance(t, Tensor):
                            st = t.t.untyped_storage()
                            if st.data_ptr() not in reg:
                                reg.add(st.data_ptr())
                                L.net_trace_region(h, st.data_ptr(), st.nbytes())
        if epochs != self._epochs:
            L.net_params_changed(h)
            self._epochs = epochs
        train = tuple(bool(m.train) for m in self.mods)
        if train != self._train:
            for m in self.mods:
                if type(m).__name__ in ("SpatialBatchNormalization", "SpatialDropout", "Dropout"):
                    L.net_set_training(h, self.ids[id(m)], int(bool(m.train)))
            self._train = train
        dp = (parallel.world_size(), parallel.sync_bn_active(), parallel.comm_handle("comm_bn"), parallel.comm_handle("comm_grad"),
              int(bool(getattr(self.root, "_bucket_overlap", False))) | (2 if parallel.hybrid() else 0))
        if dp != self._dp:
            L.net_set_dp(h, dp[0], int(dp[1]), dp[2], dp[3], int(dp[4]))
            self._dp = dp

    # ---- passes
    def _wrap(self, ptr, nd, dims, fmt):
        shape = tuple(int(dims[i]) for i in range(nd))
        ups = fmt >> 1
        n = 1
        for d in shape:
            n *= d
        n >>= 2 * ups
        t = self._view(ptr, n)
        if t is None:      # the library's own allocator (CG_NET_ALLOC=lib): wrap the raw device address
            t = torch.as_tensor(_Raw(ptr, n), device=device())
        return Tensor(t, shape, "nhwc" if fmt & 1 else "plain", ups)

    def forward(self, x):
        self._sync()
        r = rng()
        dims = (ctypes.c_long * 4)(*(list(x.shape) + [0] * (4 - len(x.shape))))
        y, ynd, yfmt, draws = ctypes.c_void_p(), ctypes.c_int(), ctypes.c_int(), ctypes.c_uint64()
        ydims = (ctypes.c_long * 4)()
        self.L.net_forward(self.h, stream(), x.ptr, len(x.shape), dims, 1 if x.fmt == "nhwc" else 0, r.seed, r.offset, r.base_ptr(),
                           ctypes.addressof(draws), ctypes.byref(y), ctypes.byref(ynd), ydims, ctypes.byref(yfmt))
        r.offset += draws.value
        self.last_draws = int(draws.value)      # static per plan: a host may reserve the stream positions behind this pass ahead of time
        self._x = x
        if self.trace:
            return Tensor(torch.empty(0), tuple(int(ydims[i]) for i in range(ynd.value)), "nhwc" if yfmt.value & 1 else "plain")
        return self._wrap(y.value, ynd.value, ydims, yfmt.value)

    def forward_pair(self, x, x2):
        """cg_net_forward_pair: forward(x) on the current stream, forward(x2) beside it on a library stream; returns pass 1's output.
        pair_join() returns pass 2's, which is also what backward() continues."""
        self._sync()
        r = rng()
        d1 = (ctypes.c_long * 4)(*(list(x.shape) + [0] * (4 - len(x.shape))))
        d2 = (ctypes.c_long * 4)(*(list(x2.shape) + [0] * (4 - len(x2.shape))))
        y, ynd, yfmt, draws = ctypes.c_void_p(), ctypes.c_int(), ctypes.c_int(), ctypes.c_uint64()
        ydims = (ctypes.c_long * 4)()
        self.L.net_forward_pair(self.h, stream(), x.ptr, len(x.shape), d1, 1 if x.fmt == "nhwc" else 0, x2.ptr, len(x2.shape), d2,
                                1 if x2.fmt == "nhwc" else 0, r.seed, r.offset, r.base_ptr(), ctypes.addressof(draws), ctypes.byref(y),
                                ctypes.byref(ynd), ydims, ctypes.byref(yfmt))
        r.offset += draws.value
        self.last_draws = int(draws.value)
        self._x = x2
        return self._wrap(y.value, ynd.value, ydims, yfmt.value)

    def pair_join(self):
        y, ynd, yfmt = ctypes.c_void_p(), ctypes.c_int(), ctypes.c_int()
        ydims = (ctypes.c_long * 4)()
        self.L.net_pair_join(self.h, stream(), ctypes.byref(y), ctypes.byref(ynd), ydims, ctypes.byref(yfmt))
        return self._wrap(y.value, ynd.value, ydims, yfmt.value)

    def backward(self, x, gy, acc, scale=1.0):
        gx, gnd, gfmt = ctypes.c_void_p(), ctypes.c_int(), ctypes.c_int()
        gdims = (ctypes.c_long * 4)()
        self.L.net_backward(self.h, stream(), x.ptr, gy.ptr, 1 if gy.fmt == "nhwc" else 0, int(bool(acc)), float(scale), ctypes.byref(gx),
                            ctypes.byref(gnd), gdims, ctypes.byref(gfmt))
        if self.trace:
            return Tensor(torch.empty(0), tuple(int(gdims[i]) for i in range(gnd.value)), "nhwc" if gfmt.value & 1 else "plain")
        return self._wrap(gx.value, gnd.value, gdims, gfmt.value)

    def set_defer_running(self, on):
        """cg_net_set_option "defer_running": the following forward passes leave the batch-norm running statistics to apply_running()."""
        self.L.net_set_option(self.h, b"defer_running", int(bool(on)))

    def apply_running(self):
        """Apply the running-statistics update of the most recent deferred forward pass, on the current stream."""
        self.L.net_apply_running(self.h, stream())

    def finish_buckets(self):
        """Join the gradient-bucket all-reduces cg_net_backward started (bucket_overlap): device-side wait, no host sync."""
        n = ctypes.c_int(0)
        self.L.net_buckets(self.h, ctypes.byref(n))
        if n.value and parallel.comm_handle("comm_grad") is not None:
            self.L.comm_wait(parallel.comm_handle("comm_grad"), stream())
        for p_ in self.pending:
            p_.finish()
        self.pending = []
        return n.value

    def module_state(self, m, which=0):
        """.output (0) / .gradInput (1) / dropout mask (2) of one module after a planned pass, or None when fused away."""
        ptr, nd, fmt = ctypes.c_void_p(), ctypes.c_int(), ctypes.c_int()
        dims = (ctypes.c_long * 4)()
        self.L.net_module_state(self.h, self.ids[id(m)], which, ctypes.byref(ptr), ctypes.byref(nd), dims, ctypes.byref(fmt))
        if not ptr.value:
            return None
        return self._wrap(ptr.value, nd.value, dims, fmt.value)

    def prologue(self):
        """The calls that rebuild this net in its present state - (entry point, argument tokens) in tools/abi_replay's syntax,
        with `H` standing for the handle - for a recorded step that starts after the net was built."""
        out = [("cg_net_create", ["H"])]
        for k in ("overlap_groups", "defer_wgrad", "winograd", "share_pool", "sampler_shared", "view_fuse", "cat_fuse", "stacking", "grouped",
                  "fusion"):
            pass   # defaults: the replaying process reads the same environment
        for parent, kind, ia, fa in self.desc:
            out.append(("cg_net_add", ["h", f"i:{parent}", f"i:{kind}", "L:" + ",".join(str(v) for v in ia), f"i:{len(ia)}",
                                       "F:" + ",".join(float(v).hex() for v in fa), f"i:{len(fa)}", "o"]))
        for mid, slot, a, b in (self._bound or []):
            out.append(("cg_net_bind", ["h", f"i:{mid}", f"i:{slot}", ("ptr", a), ("ptr", b)]))
        for m in self.mods:
            if type(m).__name__ in ("SpatialBatchNormalization", "SpatialDropout", "Dropout"):
                out.append(("cg_net_set_training", ["h", f"i:{self.ids[id(m)]}", f"i:{int(bool(m.train))}"]))
        out.append(("cg_net_params_changed", ["h"]))
        return out

    def take_trace(self):
        n = ctypes.c_size_t(0)
        self.L.net_trace_take(self.h, None, 0, ctypes.byref(n))
        buf = ctypes.create_string_buffer(n.value + 1)
        self.L.net_trace_take(self.h, ctypes.cast(buf, ctypes.c_void_p), n.value, ctypes.byref(n))
        return buf.raw[:n.value].decode()

    def stats(self):
        a, b, c, d = ctypes.c_long(), ctypes.c_long(), ctypes.c_long(), ctypes.c_size_t()
        self.L.net_stats(self.h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c), ctypes.byref(d))
        return dict(programs=a.value, launches_forward=b.value, launches_backward=c.value, bytes=d.value)
