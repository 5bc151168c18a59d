"""Device tensor handle used by the host-side module layer.

Mirrors the slice of the Torch7 tensor API the hot path touches
(SURVEY.md §8b "Tensor API used on the path"): zero/fill/copy/clone/size/
uniform/clamp/add/mul/norm.  PyTorch is used only as the allocator and
stream provider; every arithmetic method is a call through the C ABI.

Physical layouts:
  fmt 'plain' : memory order == logical (Torch7) order, row-major
  fmt 'nhwc'  : logical [N,C,H,W] stored as [N,H,W,C]  (engine-native feature maps)
  ups = 1     : logical H,W are 2x the physical ones (virtual nearest upsampling,
                consumed by the convolution's gather; nn.SpatialUpSamplingNearest)
"""
import math

import numpy as np
import torch

from . import _abi


class Epoch:
    """Mutation counter shared by all views of one flat parameter vector (drives weight re-packing)."""
    __slots__ = ("v",)

    def __init__(self):
        self.v = 0

    def bump(self):
        self.v += 1


_HAS_GPU = torch.cuda.is_available()   # constant for the process; the query itself costs ~2 us (getenv + driver probe)
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def has_gpu():
    return _HAS_GPU


def device():
    if _HAS_GPU:
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def stream():
    """hipStream_t of torch's current stream (kernels, events and collectives share it).  Called once per launch
    (~600 times per training step), hence the raw-handle fast path instead of torch.cuda.current_stream()."""
    if not _HAS_GPU:
        return 0
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def lib():
    return _abi.lib()


class SplitMix:
    """Counter-based generator, host twin of cg_rng_* (csrc/ops.hip u01): parameters are initialised on
    the host (as Torch7 does before :cuda()) from the same stream the device kernels use for masks/noise."""

    def __init__(self, seed=1):
        self.seed = int(seed)
        self.offset = 0
        self.dev_base = None  # int64 device scalar added to every kernel-side offset (hipGraph replay mode)

    def base_ptr(self):
        return self.dev_base.data_ptr() if self.dev_base is not None else None

    def enable_device_base(self):
        if self.dev_base is None:
            self.dev_base = torch.zeros(1, dtype=torch.int64, device=device())
        return self.dev_base

    def take(self, n):
        o = self.offset
        self.offset += int(n)
        return o

    def u01(self, n):
        o = self.take(n)
        with np.errstate(over="ignore"):
            i = np.arange(o + 1, o + n + 1, dtype=np.uint64)
            z = np.uint64(self.seed) + i * np.uint64(0x9E3779B97F4A7C15)
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            z = z ^ (z >> np.uint64(31))
        return (z >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / 16777216.0)

    def uniform(self, shape, lo, hi):
        n = int(np.prod(shape))
        u = self.u01(n)
        return (np.float32(lo) + (np.float32(hi) - np.float32(lo)) * u).astype(np.float32).reshape(shape)


RNG = SplitMix(1)


def manual_seed(seed):
    """torch.manualSeed / cutorch.manualSeed (train.lua:61-62,110)."""
    global RNG
    RNG = SplitMix(seed)
    return RNG


def rng():
    return RNG


class Tensor:
    __slots__ = ("t", "shape", "fmt", "ups", "epoch")

    def __init__(self, t, shape=None, fmt="plain", ups=0, epoch=None):
        self.t = t
        self.shape = tuple(int(s) for s in (shape if shape is not None else t.shape))
        self.fmt = fmt
        self.ups = ups
        self.epoch = epoch if epoch is not None else Epoch()

    # ---- construction
    @staticmethod
    def empty(shape, fmt="plain"):
        shape = tuple(int(s) for s in shape)
        return Tensor(torch.empty(math.prod(shape), dtype=torch.float32, device=device()), shape, fmt)

    @staticmethod
    def zeros(shape, fmt="plain"):
        shape = tuple(int(s) for s in shape)
        return Tensor(torch.zeros(math.prod(shape), dtype=torch.float32, device=device()), shape, fmt)

    @staticmethod
    def from_numpy(a, fmt=None):
        """Host FloatTensor -> device.  4-D arrays ([N,C,H,W]) become engine-native NHWC."""
        a = np.ascontiguousarray(a, dtype=np.float32)
        if fmt is None:
            fmt = "nhwc" if a.ndim == 4 else "plain"
        phys = np.ascontiguousarray(a.transpose(0, 2, 3, 1)) if fmt == "nhwc" else a
        t = torch.from_numpy(phys.reshape(-1)).to(device(), non_blocking=False)
        return Tensor(t, a.shape, fmt)

    # ---- introspection
    @property
    def ptr(self):
        return self.t.data_ptr()

    def nElement(self):
        return math.prod(self.shape)

    numel = nElement

    def phys_numel(self):
        return self.t.numel()

    def dim(self):
        return len(self.shape)

    def size(self, i=None):
        return self.shape if i is None else self.shape[i - 1]  # Torch7 is 1-based

    def numpy(self):
        """Device -> host in the LOGICAL (Torch7) layout."""
        a = self.t.detach().cpu().numpy()
        if self.fmt == "nhwc":
            N, Cc, H, W = self.shape
            a = a.reshape(N, H >> self.ups, W >> self.ups, Cc).transpose(0, 3, 1, 2)
            if self.ups:
                a = np.repeat(np.repeat(a, 2, axis=2), 2, axis=3)
            return np.ascontiguousarray(a)
        return a.reshape(self.shape).copy()

    def float(self):
        return self.numpy()

    # ---- views (plain only)
    def view(self, *shape):
        assert self.fmt == "plain" and self.ups == 0
        shape = tuple(int(s) for s in shape)
        assert math.prod(shape) == self.nElement()
        return Tensor(self.t, shape, "plain", 0, self.epoch)

    def rows(self, a, b):
        """t[{{a,b}}] (1-based, inclusive) of a plain tensor, or of an NHWC batch."""
        assert self.ups == 0
        per = self.nElement() // self.shape[0]
        return Tensor(self.t[(a - 1) * per:b * per], (b - a + 1,) + self.shape[1:], self.fmt, 0, self.epoch)

    # ---- mutation (all through the C ABI)
    def zero(self):
        lib().memset_zero(stream(), self.ptr, self.t.numel() * 4) if self.t.is_cuda else self.t.zero_()
        self.epoch.bump()
        return self

    def fill(self, v):
        lib().fill(stream(), self.ptr, float(v), self.t.numel()) if self.t.is_cuda else self.t.fill_(float(v))
        self.epoch.bump()
        return self

    def copy(self, src):
        """self <- src (Tensor in the same physical layout, or a host array in logical layout)."""
        if isinstance(src, Tensor):
            assert src.t.numel() == self.t.numel() and src.fmt == self.fmt and src.ups == self.ups
            if self.t.is_cuda:
                lib().memcpy_d2d(stream(), self.ptr, src.ptr, self.t.numel() * 4)
            else:
                self.t.copy_(src.t)
        else:
            a = np.ascontiguousarray(src, dtype=np.float32).reshape(self.shape)
            if self.fmt == "nhwc":
                a = np.ascontiguousarray(a.transpose(0, 2, 3, 1))
            self.t.copy_(torch.from_numpy(a.reshape(-1)))
        self.epoch.bump()
        return self

    def clone(self):
        out = Tensor(torch.empty_like(self.t), self.shape, self.fmt, self.ups)
        return out.copy(self)

    def mul(self, a):
        lib().scale(stream(), self.ptr, float(a), self.t.numel())
        self.epoch.bump()
        return self

    def add(self, alpha, other=None):
        """t:add(alpha, other): self += alpha*other ; t:add(other): self += other."""
        if other is None:
            alpha, other = 1.0, alpha
        lib().axpy(stream(), float(alpha), other.ptr, self.ptr, self.t.numel())
        self.epoch.bump()
        return self

    def clamp(self, lo, hi):
        lib().clamp(stream(), self.ptr, float(lo), float(hi), self.t.numel())
        self.epoch.bump()
        return self

    def norm(self, p=2):
        """torch.norm(t, p) for p in {1,2} (adversarial.lua:94-95); synchronises."""
        acc = torch.zeros(1, dtype=torch.float64, device=self.t.device)
        if p == 1:
            lib().sumabs(stream(), self.ptr, self.t.numel(), acc.data_ptr())
            return float(acc.item())
        lib().sumsq(stream(), self.ptr, self.t.numel(), acc.data_ptr())
        return float(np.sqrt(acc.item()))

    def uniform(self, lo, hi):
        """Host-side initialisation from the shared counter stream (see SplitMix)."""
        self.copy(rng().uniform(self.shape, lo, hi))
        return self

    def __repr__(self):
        return f"catgan.Tensor{self.shape}[{self.fmt}{'+ups' if self.ups else ''}]"


class Workspace:
    """Grow-only device scratch for the GEMM entry points (split-K partials), one per HIP stream so that work
    forked onto a side stream never shares partials with the main stream."""

    def __init__(self):
        self.per_stream = {}

    def get(self, nbytes):
        nbytes = max(int(nbytes), 4096)
        key = stream()
        t = self.per_stream.get(key)
        if t is None or t.numel() < nbytes:
            t = torch.empty(int(nbytes * 1.25), dtype=torch.uint8, device=device())
            self.per_stream[key] = t
        return t.data_ptr(), t.numel()


WS = Workspace()
