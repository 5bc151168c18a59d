"""
oracle/oracle.py — CPU restatement of the reference's G+D step (TEST INFRASTRUCTURE).

PARITY UNPINNED: aleju/cat-generator ships no tests / golden vectors and its
arithmetic lives in un-vendored Torch7 rocks (see oracle/ops.c header).  This
module restates, in Torch7's own NCHW layouts, the module graph of
models.lua (G32up-c :196-228, G32up :138-160, D32_st3 :640-711, spatial
transformer :814-906), the in-tree LeakyReLU (LeakyReLU.lua:13-31), the step
semantics of adversarial.lua:72-167 / :171-215 / :221-266 and Torch7's
optim.adam.  Heavy operators (conv / linear / bilinear sampler) are in
oracle/ops.c; everything elementwise is numpy fp32.

ASSUMPTIONS.  Everything the reference takes from an un-vendored rock is RECALLED, not read; where the
rock changed over time the oracle has to pick an era (the reference's README dates it to late 2015 / early 2016, "cudnn3").
"in-tree" = the reference's own file says so; "recalled" = upstream Torch7 behaviour of that era as we remember it.  The
last column names the test that pins the choice INSIDE this repo (engine == oracle) and would have to change with it; no
test can pin it against Torch7 itself - that is what "parity unpinned" means.

| # | choice made here | source | the other era / reading | what would change | pinned by |
|---|---|---|---|---|---|
| 1 | nn.LeakyReLU(): slope 0.333, x == 0 takes the positive branch | in-tree, LeakyReLU.lua:8,13-31 | LeakyReLU.lua:2-4 RETURNS EARLY when upstream nn already defines nn.LeakyReLU (torch/nn gained one around Dec 2015, default negval 1/100, x == 0 on the negative branch): models.lua:845-853 call nn.LeakyReLU() without an argument, so on such an install the localisation nets would run with slope 0.01 | the 12 LeakyReLUs of the four localisation nets (models.lua:845-853): the transformers' theta and D's gradient; nothing in G | LeakyReLU.forward/backward below; tests/test_gpu_parity.py::test_activations_and_bce, test_spatial_transformer_module |
| 2 | nn.SpatialDropout(p): train y = x * mask, mask ~ Bernoulli(1 - p) per (sample, plane), NO 1/(1-p) rescale; evaluate y = (1 - p) x; default p = 0.5 (models.lua:695 passes none) | recalled (nn/SpatialDropout.lua of that era) | later nn versions added a `stochasticInference` flag but never the rescale; a PyTorch-style dropout2d rescales by 1/(1-p) in training | D's activations behind the six SpatialDropouts by 1/(1-p) = 1.25 / 2; D's gradient likewise | SpatialDropout below; tests/test_oracle_vs_torch.py::test_whole_step_gradients_match_torch_autograd (explicit masks), test_gpu_parity.py::test_dropout_masks_share_the_counter_stream |
| 3 | nn.Dropout(): p = 0.5, "v2": train y = x * mask / (1 - p), evaluate identity | recalled (nn/Dropout.lua, v1 = false default) | v1 (nn.Dropout(p, true)): no rescale in training, y = (1 - p) x in evaluate | the head's Linear(256,1) input by a factor 2 (models.lua:699) | Dropout below; tests/test_gpu_parity.py::test_concat_dropout_and_head_launches |
| 4 | nn.SpatialBatchNormalization(n): eps 1e-5, momentum 0.1, affine, gamma ~ U(0,1), beta 0, normalise with the BIASED batch variance, running_var updated with the unbiased one | recalled (nn/BatchNormalization.lua: weight:uniform()) | later nn initialises gamma = 1 (as PyTorch does); cudnn.SpatialBatchNormalization has a different eps floor | G's initial parameters (the three BN gammas) and with them every synthetic benchmark's numerics, not the kernels | SpatialBatchNormalization below; tests/test_abi_and_host.py::test_models_match_oracle_structure_bit_for_bit, test_oracle_vs_torch.py::test_bn_train_matches_torch |
| 5 | optim.adam: m, v updates, then x -= lr * sqrt(1 - b2^t) / (1 - b1^t) * m / (sqrt(v) + eps): eps added to sqrt(v) WITHOUT bias-correcting v; t starts at 0 and is incremented before use; defaults 1e-3, 0.9, 0.999, 1e-8 | recalled (optim/adam.lua) | PyTorch: eps added after dividing sqrt(v) by sqrt(1 - b2^t) | every update with |g| near eps; first steps visibly (closed forms for both in the test) | adam() below; tests/test_oracle_vs_torch.py::test_torch7_adam_five_steps_against_a_scalar_double_loop, test_bce_and_adam_known_answers, test_gpu_parity.py::test_adam_and_fused_penalty_clamp |
| 6 | nn.BCECriterion: eps = 1e-12 inside both logs, mean over the batch (sizeAverage) | recalled (nn/BCECriterion.lua) | PyTorch clamps log at -100 instead; an older nn had no eps | only saturated D outputs (|logit| > ~27) | BCECriterion below; tests/test_gpu_parity.py::test_activations_and_bce |
| 7 | image.scale (dataset.lua:129-131): separable, on the float image; down-scaling = box average with fractional ends, up-scaling = linear with the last sample copied | recalled (image/generic/image.c scaleLinear_rowcol) | 'simple' / 'bicubic' modes exist upstream but are not the default the reference uses | the real half of every D batch (input pipeline, row f2), nothing in the timed step (the pool is resident) | tests/test_dataset_cli.py::test_image_scale_follows_the_oracle_rule, test_oracle_vs_torch.py::test_image_scale_rule_against_torch_interpolate |
| 8 | stn: AffineTransformMatrixGenerator consumes [theta][scale][tx, ty] in that order, AffineGridGeneratorBHWD puts y first, BilinearSamplerBHWD samples corner-aligned with zeros outside; theta = +pi/2 turns the picture clockwise | recalled (qassemoquab/stnbhwd) + the in-tree initial bias {0, 1, 0, 0} (models.lua:859-860) which fixes the order rot, scale, trans | the rotation sign and the (y, x) order are conventions of the rock | the learned transformer parameters' meaning; at the identity initialisation nothing | tests/test_oracle_vs_torch.py::test_transformer_conventions_as_geometry, test_transformer_trio_second_restatement_with_explicit_axis_swap; test_gpu_parity.py::test_spatial_transformer_module |
| 9 | weight-init 'heuristic' is NOT recursive (weight-init.lua:52 iterates net.modules): nn.Concat's children keep the default reset(); cudnn.SpatialConvolution does not match the typename test but gets its bias zeroed (:70-72) | in-tree, weight-init.lua:14-16,52-72 | - | - | tests/test_abi_and_host.py::test_models_match_oracle_structure_bit_for_bit (Concat children keep their default bias) |
| 10 | adversarial.lua:206 multiplies the L1 sign term by G_L2 (an upstream slip, inert at the defaults) and is kept | in-tree | - | - | Trainer.step below |

Every [upstream] semantic choice is also listed in SURVEY.md Appendix B.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module.  The product package never does.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle_ops.so")
_lib = None

f32 = np.float32


def build(force=False):
    """Compile oracle/ops.c with gcc (AVX2+FMA baseline so the .so runs on any x86-64-v3 host)."""
    src = os.path.join(_HERE, "ops.c")
    if not force and os.path.exists(_LIB_PATH) and os.path.getmtime(_LIB_PATH) >= os.path.getmtime(src):
        return _LIB_PATH
    cmd = ["gcc", "-O3", "-march=x86-64-v3", "-fopenmp", "-fPIC", "-shared", "-std=gnu11", src, "-o", _LIB_PATH, "-lm"]
    subprocess.check_call(cmd)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_num_threads.restype = C.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _c(a):
    return np.ascontiguousarray(a, dtype=f32)


def num_threads():
    return lib().orc_num_threads()


def set_num_threads(n):
    lib().orc_set_num_threads(int(n))


# ----------------------------------------------------------------------------- RNG
def u01(n, seed, offset):
    out = np.empty(int(n), f32)
    lib().orc_rng_u01(_p(out), C.c_long(int(n)), C.c_uint64(seed), C.c_uint64(offset))
    return out


class RNG:
    """Counter-based stream shared (by construction) with the engine's cg_rng_* kernels."""

    def __init__(self, seed=1):
        self.seed = int(seed)
        self.offset = 0

    def take(self, n):
        o = self.offset
        self.offset += int(n)
        return o

    def uniform(self, shape, lo, hi):
        n = int(np.prod(shape))
        u = u01(n, self.seed, self.take(n))
        return (f32(lo) + (f32(hi) - f32(lo)) * u).astype(f32).reshape(shape)

    def bernoulli(self, shape, keep, value=1.0):
        n = int(np.prod(shape))
        u = u01(n, self.seed, self.take(n))
        return np.where(u < f32(keep), f32(value), f32(0)).astype(f32).reshape(shape)


# ------------------------------------------------------------------- heavy ops (C)
def conv2d_forward(x, w, b, pad):
    x, w = _c(x), _c(w)
    N, Cin, H, W = x.shape
    Cout, _, kH, kW = w.shape
    Ho, Wo = H + 2 * pad - kH + 1, W + 2 * pad - kW + 1
    y = np.empty((N, Cout, Ho, Wo), f32)
    lib().orc_conv2d_forward(_p(x), _p(w), _p(_c(b)) if b is not None else None, _p(y), N, Cin, H, W, Cout, kH, kW, pad, pad)
    return y


def conv2d_backward_data(dy, w, in_shape, pad):
    dy, w = _c(dy), _c(w)
    N, Cin, H, W = in_shape
    Cout, _, kH, kW = w.shape
    dx = np.empty(in_shape, f32)
    lib().orc_conv2d_backward_data(_p(dy), _p(w), _p(dx), N, Cin, H, W, Cout, kH, kW, pad, pad)
    return dx


def conv2d_backward_weight(x, dy, gw, gb, pad, scale=1.0):
    x, dy = _c(x), _c(dy)
    N, Cin, H, W = x.shape
    Cout, _, kH, kW = gw.shape
    assert gw.flags.c_contiguous and gw.dtype == f32
    lib().orc_conv2d_backward_weight(_p(x), _p(dy), _p(gw), _p(gb), N, Cin, H, W, Cout, kH, kW, pad, pad, C.c_float(scale))


def linear_forward(x, w, b):
    x, w = _c(x), _c(w)
    N, i = x.shape
    o = w.shape[0]
    y = np.empty((N, o), f32)
    lib().orc_linear_forward(_p(x), _p(w), _p(_c(b)) if b is not None else None, _p(y), N, i, o)
    return y


def linear_backward_data(dy, w):
    dy, w = _c(dy), _c(w)
    N, o = dy.shape
    i = w.shape[1]
    dx = np.empty((N, i), f32)
    lib().orc_linear_backward_data(_p(dy), _p(w), _p(dx), N, i, o)
    return dx


def linear_backward_weight(x, dy, gw, gb, scale=1.0):
    x, dy = _c(x), _c(dy)
    N, i = x.shape
    o = dy.shape[1]
    assert gw.flags.c_contiguous and gw.dtype == f32
    lib().orc_linear_backward_weight(_p(x), _p(dy), _p(gw), _p(gb), N, i, o, C.c_float(scale))


def bilinear_forward(img, grid):
    img, grid = _c(img), _c(grid)
    N, Hi, Wi, Cc = img.shape
    _, Ho, Wo, _ = grid.shape
    out = np.empty((N, Ho, Wo, Cc), f32)
    lib().orc_bilinear_forward(_p(img), _p(grid), _p(out), N, Hi, Wi, Cc, Ho, Wo)
    return out


def bilinear_backward(img, grid, gout):
    img, grid, gout = _c(img), _c(grid), _c(gout)
    N, Hi, Wi, Cc = img.shape
    _, Ho, Wo, _ = grid.shape
    gimg = np.empty_like(img)
    ggrid = np.empty_like(grid)
    lib().orc_bilinear_backward(_p(img), _p(grid), _p(gout), _p(gimg), _p(ggrid), N, Hi, Wi, Cc, Ho, Wo)
    return gimg, ggrid


# ------------------------------------------------------------- module graph (nn)
class Module:
    """Torch7 nn.Module protocol restated: forward/backward with accGradParameters (scale 1)."""

    def __init__(self):
        self.params = []  # list of (name) attributes holding parameter arrays, in Torch order
        self.train = True

    def parameters(self):
        return [(getattr(self, n), getattr(self, "grad_" + n)) for n in self.params]

    def modules(self):
        return [self]

    def training(self, flag=True):
        for m in self.modules():
            m.train = flag


class Linear(Module):
    """nn.Linear(in,out): y = x W^T + b; default reset U(+-1/sqrt(in)) [upstream]."""

    def __init__(self, i, o, rng):
        super().__init__()
        stdv = 1.0 / np.sqrt(i)
        self.weight = rng.uniform((o, i), -stdv, stdv)
        self.bias = rng.uniform((o,), -stdv, stdv)
        self.grad_weight = np.zeros_like(self.weight)
        self.grad_bias = np.zeros_like(self.bias)
        self.params = ["weight", "bias"]
        self.typename = "nn.Linear"

    def reset(self, stdv, rng):
        s = stdv * np.sqrt(3.0)  # nn.Linear:reset(stdv) -> stdv*sqrt(3) [upstream]
        self.weight[...] = rng.uniform(self.weight.shape, -s, s)
        self.bias[...] = rng.uniform(self.bias.shape, -s, s)

    def fan_in(self):
        return self.weight.shape[1]

    def forward(self, x):
        self.x = x
        return linear_forward(x, self.weight, self.bias)

    def backward(self, dy, need_input_grad=True):
        linear_backward_weight(self.x, dy, self.grad_weight, self.grad_bias)
        return linear_backward_data(dy, self.weight) if need_input_grad else None


class Conv(Module):
    """nn.SpatialConvolution / cudnn.SpatialConvolution (stride 1); default reset U(+-1/sqrt(kW*kH*nIn))."""

    def __init__(self, nin, nout, k, pad, rng, typename="nn.SpatialConvolution"):
        super().__init__()
        stdv = 1.0 / np.sqrt(k * k * nin)
        self.weight = rng.uniform((nout, nin, k, k), -stdv, stdv)
        self.bias = rng.uniform((nout,), -stdv, stdv)
        self.grad_weight = np.zeros_like(self.weight)
        self.grad_bias = np.zeros_like(self.bias)
        self.pad = pad
        self.params = ["weight", "bias"]
        self.typename = typename

    def reset(self, stdv, rng):
        s = stdv * np.sqrt(3.0)
        self.weight[...] = rng.uniform(self.weight.shape, -s, s)
        self.bias[...] = rng.uniform(self.bias.shape, -s, s)

    def fan_in(self):
        return int(np.prod(self.weight.shape[1:]))

    def forward(self, x):
        self.x = x
        return conv2d_forward(x, self.weight, self.bias, self.pad)

    def backward(self, dy, need_input_grad=True):
        conv2d_backward_weight(self.x, dy, self.grad_weight, self.grad_bias, self.pad)
        return conv2d_backward_data(dy, self.weight, self.x.shape, self.pad) if need_input_grad else None


class PReLU(Module):
    """nn.PReLU(): one shared slope, init 0.25; dalpha = sum_{x<=0} x*dy [upstream]."""

    def __init__(self):
        super().__init__()
        self.weight = np.full((1,), 0.25, f32)
        self.grad_weight = np.zeros((1,), f32)
        self.params = ["weight"]

    def forward(self, x):
        self.x = x
        return np.where(x > 0, x, self.weight[0] * x).astype(f32)

    def backward(self, dy):
        x = self.x
        neg = x <= 0
        self.grad_weight[0] += f32(np.sum((x * dy)[neg], dtype=np.float64))
        return np.where(x > 0, dy, self.weight[0] * dy).astype(f32)


class LeakyReLU(Module):
    """LeakyReLU.lua:13-31: slope s=0.333 for x<0, 1 for x>=0 (x==0 takes the positive branch)."""

    def __init__(self, s=0.333):
        super().__init__()
        self.s = f32(s)

    def forward(self, x):
        self.x = x
        return np.where(x >= 0, x, self.s * x).astype(f32)

    def backward(self, dy):
        return np.where(self.x >= 0, dy, self.s * dy).astype(f32)


class Sigmoid(Module):
    def forward(self, x):
        self.y = (1.0 / (1.0 + np.exp(-x.astype(f32)))).astype(f32)
        return self.y

    def backward(self, dy):
        return (dy * (1.0 - self.y) * self.y).astype(f32)


class View(Module):
    def __init__(self, *shape):
        super().__init__()
        self.shape = shape

    def forward(self, x):
        self.in_shape = x.shape
        return np.ascontiguousarray(x).reshape((x.shape[0],) + tuple(self.shape))

    def backward(self, dy):
        return np.ascontiguousarray(dy).reshape(self.in_shape)


class UpSample2(Module):
    """nn.SpatialUpSamplingNearest(2)."""

    def forward(self, x):
        return np.repeat(np.repeat(x, 2, axis=2), 2, axis=3)

    def backward(self, dy):
        N, Cc, H, W = dy.shape
        d = dy.reshape(N, Cc, H // 2, 2, W // 2, 2)
        return ((d[:, :, :, 0, :, 0] + d[:, :, :, 0, :, 1]) + (d[:, :, :, 1, :, 0] + d[:, :, :, 1, :, 1])).astype(f32)


class SBN(Module):
    """nn.SpatialBatchNormalization(n): eps 1e-5, momentum 0.1, gamma~U(0,1), beta 0 [upstream]."""

    def __init__(self, n, rng):
        super().__init__()
        self.weight = rng.uniform((n,), 0.0, 1.0)
        self.bias = np.zeros((n,), f32)
        self.grad_weight = np.zeros((n,), f32)
        self.grad_bias = np.zeros((n,), f32)
        self.running_mean = np.zeros((n,), f32)
        self.running_var = np.ones((n,), f32)
        self.eps, self.momentum = 1e-5, 0.1
        self.params = ["weight", "bias"]
        # hooks for data-parallel tests: all-reduce of the (sum, sumsq) statistics
        self.stat_allreduce = None

    def forward(self, x):
        self.x = x
        N, Cc, H, W = x.shape
        if not self.train:
            inv = 1.0 / np.sqrt(self.running_var + f32(self.eps))
            return ((x - self.running_mean[None, :, None, None]) * inv[None, :, None, None] * self.weight[None, :, None, None]
                    + self.bias[None, :, None, None]).astype(f32)
        cnt = float(N * H * W)
        s1 = x.sum(axis=(0, 2, 3), dtype=np.float64)
        s2 = (x.astype(np.float64) ** 2).sum(axis=(0, 2, 3))
        if self.stat_allreduce is not None:
            s1, s2, cnt = self.stat_allreduce(s1, s2, cnt)
        mean = s1 / cnt
        var = np.maximum(s2 / cnt - mean * mean, 0.0)
        self.count = cnt
        self.save_mean = mean.astype(f32)
        self.save_invstd = (1.0 / np.sqrt(var + self.eps)).astype(f32)
        self.running_mean = ((1 - self.momentum) * self.running_mean + self.momentum * mean).astype(f32)
        unb = var * cnt / (cnt - 1.0) if cnt > 1 else var
        self.running_var = ((1 - self.momentum) * self.running_var + self.momentum * unb).astype(f32)
        m, i = self.save_mean[None, :, None, None], self.save_invstd[None, :, None, None]
        return ((x - m) * i * self.weight[None, :, None, None] + self.bias[None, :, None, None]).astype(f32)

    def backward(self, dy):
        x = self.x
        m, i = self.save_mean[None, :, None, None], self.save_invstd[None, :, None, None]
        xh = ((x - m) * i).astype(f32)
        s1 = dy.sum(axis=(0, 2, 3), dtype=np.float64)
        s2 = (dy * xh).sum(axis=(0, 2, 3), dtype=np.float64)
        self.grad_bias += s1.astype(f32)
        self.grad_weight += s2.astype(f32)
        cnt = self.count
        if self.stat_allreduce is not None:
            s1, s2, _ = self.stat_allreduce(s1, s2, 0.0)
        m1 = (s1 / cnt).astype(f32)[None, :, None, None]
        m2 = (s2 / cnt).astype(f32)[None, :, None, None]
        return (self.weight[None, :, None, None] * i * (dy - m1 - xh * m2)).astype(f32)


class AvgPool2(Module):
    def forward(self, x):
        N, Cc, H, W = x.shape
        d = x.reshape(N, Cc, H // 2, 2, W // 2, 2)
        return ((d[:, :, :, 0, :, 0] + d[:, :, :, 0, :, 1] + d[:, :, :, 1, :, 0] + d[:, :, :, 1, :, 1]) * f32(0.25)).astype(f32)

    def backward(self, dy):
        return (np.repeat(np.repeat(dy, 2, axis=2), 2, axis=3) * f32(0.25)).astype(f32)


class MaxPool2(Module):
    """nn.SpatialMaxPooling(2,2): gradient to the first max in scan order."""

    def forward(self, x):
        N, Cc, H, W = x.shape
        d = x.reshape(N, Cc, H // 2, 2, W // 2, 2)
        cand = np.stack([d[:, :, :, 0, :, 0], d[:, :, :, 0, :, 1], d[:, :, :, 1, :, 0], d[:, :, :, 1, :, 1]], axis=-1)
        self.arg = np.argmax(cand, axis=-1)
        self.in_shape = x.shape
        return np.max(cand, axis=-1).astype(f32)

    def backward(self, dy):
        N, Cc, H, W = self.in_shape
        dx = np.zeros((N, Cc, H // 2, 2, W // 2, 2), f32)
        for a, (iy, ix) in enumerate([(0, 0), (0, 1), (1, 0), (1, 1)]):
            dx[:, :, :, iy, :, ix] = np.where(self.arg == a, dy, 0)
        return dx.reshape(N, Cc, H, W)


class SpatialDropout(Module):
    """nn.SpatialDropout(p): train y = x*mask[n,c], NO rescale; eval y = (1-p)*x [upstream, era]."""

    def __init__(self, p, rng):
        super().__init__()
        self.p, self.rng, self.fixed = p, rng, None

    def forward(self, x):
        if not self.train:
            return (x * f32(1 - self.p)).astype(f32)
        N, Cc = x.shape[:2]
        self.mask = self.fixed if self.fixed is not None else self.rng.bernoulli((N, Cc), 1 - self.p, 1.0)
        return (x * self.mask[:, :, None, None]).astype(f32)

    def backward(self, dy):
        if not self.train:
            return (dy * f32(1 - self.p)).astype(f32)
        return (dy * self.mask[:, :, None, None]).astype(f32)


class Dropout(Module):
    """nn.Dropout() v2: train y = x*mask/(1-p); eval identity [upstream]."""

    def __init__(self, p, rng):
        super().__init__()
        self.p, self.rng, self.fixed = p, rng, None

    def forward(self, x):
        if not self.train:
            return x
        self.mask = self.fixed if self.fixed is not None else self.rng.bernoulli(x.shape, 1 - self.p, 1.0 / (1 - self.p))
        return (x * self.mask).astype(f32)

    def backward(self, dy):
        return (dy * self.mask).astype(f32) if self.train else dy


class ToBHWD(Module):
    """nn.Transpose({3,4},{2,4}) (models.lua:870): NCHW -> BHWD."""

    def forward(self, x):
        return np.ascontiguousarray(x.transpose(0, 2, 3, 1))

    def backward(self, dy):
        return np.ascontiguousarray(dy.transpose(0, 3, 1, 2))


class FromBHWD(Module):
    """nn.Transpose({2,4},{3,4}) (models.lua:903): BHWD -> NCHW."""

    def forward(self, x):
        return np.ascontiguousarray(x.transpose(0, 3, 1, 2))

    def backward(self, dy):
        return np.ascontiguousarray(dy.transpose(0, 2, 3, 1))


class AffineMatrix(Module):
    """nn.AffineTransformMatrixGenerator(rot,scale,trans) [stn]: T = R*S*Tr, top 2 rows, acts on (y,x,1)."""

    def __init__(self, rot, scale, trans):
        super().__init__()
        self.rot, self.scale, self.trans = rot, scale, trans

    def _unpack(self, p):
        N = p.shape[0]
        k = 0
        th = np.zeros(N, f32); sc = np.ones(N, f32); tx = np.zeros(N, f32); ty = np.zeros(N, f32)
        if self.rot:
            th = p[:, k]; k += 1
        if self.scale:
            sc = p[:, k]; k += 1
        if self.trans:
            tx, ty = p[:, k], p[:, k + 1]
        return th, sc, tx, ty

    def forward(self, p):
        self.p = p
        th, sc, tx, ty = self._unpack(p)
        c, s = np.cos(th).astype(f32), np.sin(th).astype(f32)
        T = np.empty((p.shape[0], 2, 3), f32)
        T[:, 0, 0] = c * sc; T[:, 0, 1] = -s * sc; T[:, 0, 2] = c * sc * tx - s * sc * ty
        T[:, 1, 0] = s * sc; T[:, 1, 1] = c * sc; T[:, 1, 2] = s * sc * tx + c * sc * ty
        return T

    def backward(self, gT):
        th, sc, tx, ty = self._unpack(self.p)
        c, s = np.cos(th).astype(f32), np.sin(th).astype(f32)
        g = gT.reshape(-1, 6)
        cols = []
        if self.rot:
            cols.append(g[:, 0] * (-s * sc) + g[:, 1] * (-c * sc) + g[:, 2] * (-s * sc * tx - c * sc * ty)
                        + g[:, 3] * (c * sc) + g[:, 4] * (-s * sc) + g[:, 5] * (c * sc * tx - s * sc * ty))
        if self.scale:
            cols.append(g[:, 0] * c + g[:, 1] * (-s) + g[:, 2] * (c * tx - s * ty)
                        + g[:, 3] * s + g[:, 4] * c + g[:, 5] * (s * tx + c * ty))
        if self.trans:
            cols.append(g[:, 2] * (c * sc) + g[:, 5] * (s * sc))
            cols.append(g[:, 2] * (-s * sc) + g[:, 5] * (c * sc))
        return np.stack(cols, axis=1).astype(f32)


class AffineGrid(Module):
    """nn.AffineGridGeneratorBHWD(H,W) [stn]: grid[b,i,j,:] = T_b (y_i, x_j, 1), y_i=-1+2i/(H-1)."""

    def __init__(self, H, W):
        super().__init__()
        ys = (-1 + 2 * np.arange(H, dtype=f32) / f32(H - 1)).astype(f32)
        xs = (-1 + 2 * np.arange(W, dtype=f32) / f32(W - 1)).astype(f32)
        base = np.ones((H, W, 3), f32)
        base[:, :, 0] = ys[:, None]
        base[:, :, 1] = xs[None, :]
        self.base = base.reshape(H * W, 3)
        self.H, self.W = H, W

    def forward(self, T):
        N = T.shape[0]
        g = np.einsum("pk,nrk->npr", self.base, T).astype(f32)
        return g.reshape(N, self.H, self.W, 2)

    def backward(self, gg):
        N = gg.shape[0]
        return np.einsum("npr,pk->nrk", gg.reshape(N, -1, 2).astype(np.float64), self.base.astype(np.float64)).astype(f32)


class Sequential(Module):
    def __init__(self, *mods):
        super().__init__()
        self.mods = list(mods)

    def add(self, m):
        self.mods.append(m)
        return self

    def modules(self):
        out = [self]
        for m in self.mods:
            out += m.modules()
        return out

    def parameters(self):
        out = []
        for m in self.mods:
            out += m.parameters()
        return out

    def forward(self, x):
        for m in self.mods:
            x = m.forward(x)
        return x

    def backward(self, dy):
        for m in reversed(self.mods):
            dy = m.backward(dy)
        return dy


class Concat(Sequential):
    """nn.Concat(2): every branch sees the input; outputs concatenated on channels; bwd sums gradInputs."""

    def forward(self, x):
        outs = [m.forward(x) for m in self.mods]
        self.sizes = [o.shape[1] for o in outs]
        return np.concatenate(outs, axis=1)

    def backward(self, dy):
        off, gi = 0, None
        for m, s in zip(self.mods, self.sizes):
            g = m.backward(np.ascontiguousarray(dy[:, off:off + s]))
            gi = g if gi is None else (gi + g).astype(f32)
            off += s
        return gi


class SpatialTransformer(Module):
    """createSpatialTransformer (models.lua:814-906): ConcatTable{Transpose, locnet->ATMG->AGG} ->
    BilinearSamplerBHWD -> Transpose back."""

    def __init__(self, rot, scale, trans, size, channels, rng):
        super().__init__()
        init_bias = []
        if rot: init_bias += [0.0]
        if scale: init_bias += [1.0]
        if trans: init_bias += [0.0, 0.0]
        nparams = len(init_bias)
        nh = size // 4
        net = Sequential(AvgPool2(), Conv(channels, 16, 3, 1, rng), LeakyReLU(), Conv(16, 16, 3, 1, rng), LeakyReLU(),
                         AvgPool2(), View(16 * nh * nh), Linear(16 * nh * nh, 64, rng), LeakyReLU())
        classifier = Linear(64, nparams, rng)
        net.add(classifier)
        weight_init_heuristic(net, rng)  # models.lua:857
        classifier.weight[...] = 0  # models.lua:859
        classifier.bias[...] = np.asarray(init_bias, f32)  # models.lua:860
        self.loc = net
        self.atm = AffineMatrix(rot, scale, trans)
        self.agg = AffineGrid(size, size)
        self.to_bhwd, self.from_bhwd = ToBHWD(), FromBHWD()

    def modules(self):
        return [self] + self.loc.modules()

    def parameters(self):
        return self.loc.parameters()

    def forward(self, x):
        self.img = self.to_bhwd.forward(x)
        self.grid = self.agg.forward(self.atm.forward(self.loc.forward(x)))
        return self.from_bhwd.forward(bilinear_forward(self.img, self.grid))

    def backward(self, dy):
        gimg, ggrid = bilinear_backward(self.img, self.grid, self.from_bhwd.backward(dy))
        g1 = self.to_bhwd.backward(gimg)
        g2 = self.loc.backward(self.atm.backward(self.agg.backward(ggrid)))
        return (g1 + g2).astype(f32)  # ConcatTable sums the branch gradInputs


def weight_init_heuristic(net, rng):
    """weight-init.lua:14-16,52-72 'heuristic': only TOP-LEVEL modules; nn.SpatialConvolution / nn.Linear get
    reset(sqrt(1/(3*fan_in))); every top-level module with a bias gets it zeroed."""
    for m in net.mods:
        tn = getattr(m, "typename", None)
        if tn in ("nn.SpatialConvolution", "nn.Linear"):
            m.reset(np.sqrt(1.0 / (3.0 * m.fan_in())), rng)
        if hasattr(m, "bias") and isinstance(getattr(m, "bias", None), np.ndarray):
            m.bias[...] = 0
    return net


# ------------------------------------------------------------------ model factories
def create_G32up_c(channels, noise_dim, rng, base=4):
    """models.lua:196-228 create_G_decoder_upsampling32c (base=8: the 64x64 extension of BASELINE config #5)."""
    cd = "cudnn.SpatialConvolution"  # does not match weight-init's typename test (weight-init.lua:54)
    m = Sequential(
        Linear(noise_dim, 512 * base * base, rng), PReLU(), View(512, base, base),
        UpSample2(), Conv(512, 512, 3, 1, rng, cd), SBN(512, rng), PReLU(),
        UpSample2(), Conv(512, 256, 3, 1, rng, cd), SBN(256, rng), PReLU(),
        UpSample2(), Conv(256, 128, 5, 2, rng, cd), SBN(128, rng), PReLU(),
        Conv(128, channels, 3, 1, rng, cd), Sigmoid())
    return weight_init_heuristic(m, rng)


def create_G32up(channels, noise_dim, rng):
    """models.lua:138-160 create_G_decoder_upsampling32."""
    cd = "cudnn.SpatialConvolution"
    m = Sequential(
        Linear(noise_dim, 128 * 8 * 8, rng), View(128, 8, 8), PReLU(),
        UpSample2(), Conv(128, 256, 5, 2, rng, cd), SBN(256, rng), PReLU(),
        UpSample2(), Conv(256, 128, 5, 2, rng, cd), SBN(128, rng), PReLU(),
        Conv(128, channels, 3, 1, rng, cd), Sigmoid())
    return weight_init_heuristic(m, rng)


def create_D32_st3(channels, size, rng):
    """models.lua:640-711 create_D32_st3 (CPU form, no nn.Copy)."""
    conv = Sequential()
    conv.add(SpatialTransformer(True, False, False, size, channels, rng))
    conv.add(Conv(channels, 64, 3, 1, rng)); conv.add(PReLU())
    conv.add(Conv(64, 64, 3, 1, rng)); conv.add(PReLU())
    conv.add(AvgPool2()); conv.add(SpatialDropout(0.2, rng))
    concy = Concat()
    for _ in range(3):
        b = Sequential(SpatialTransformer(True, True, True, size // 2, 64, rng),
                       Conv(64, 64, 3, 1, rng), PReLU(), MaxPool2(), SpatialDropout(0.2, rng),
                       Conv(64, 64, 3, 1, rng), PReLU())
        concy.add(b)
    concy.add(Sequential(Conv(64, 128, 5, 2, rng), PReLU(), MaxPool2(), SpatialDropout(0.2, rng),
                         Conv(128, 128, 7, 3, rng), PReLU()))
    conv.add(concy)
    conv.add(SpatialDropout(0.5, rng))
    feat = 320 * (size // 4) * (size // 4)
    conv.add(View(feat))
    conv.add(Linear(feat, 256, rng)); conv.add(PReLU()); conv.add(Dropout(0.5, rng))
    conv.add(Linear(256, 1, rng)); conv.add(Sigmoid())
    return weight_init_heuristic(conv, rng)  # models.lua:708 (top level only; Concat children keep defaults)


def get_parameters(net):
    """Module:getParameters(): flatten depth-first, weight then bias; re-point module params to views."""
    plist = net.parameters()
    n = sum(p.size for p, _ in plist)
    flat, gflat = np.zeros(n, f32), np.zeros(n, f32)
    off = 0
    owners = []
    for m in net.modules():
        for name in m.params:
            owners.append((m, name))
    assert len(owners) == len(plist)
    for (m, name) in owners:
        p = getattr(m, name)
        sz = p.size
        flat[off:off + sz] = p.reshape(-1)
        setattr(m, name, flat[off:off + sz].reshape(p.shape))
        setattr(m, "grad_" + name, gflat[off:off + sz].reshape(p.shape))
        off += sz
    return flat, gflat


# ---------------------------------------------------------------- input pipeline (dataset.lua:123-131,166)
def image_scale(img, w, h):
    """image.scale(img, w, h), default mode 'bilinear', of the torch `image` rock [upstream, RECALLED - the rock is not in the
    reference tree; assumption stated here and in DESIGN.md]: generic/image.c scales every row to the target width into a float
    temporary, then every column to the target height (scaleLinear_rowcol).  Along one axis (source length Ls, target Ld):
      Ld == Ls  copy
      Ld >  Ls  scale = (float)(Ls-1)/(Ld-1); for d < Ld-1: s = d*scale, i = (long)s, f = s - i, out = (1-f) src[i] + f src[i+1];
                out[Ld-1] = src[Ls-1]
      Ld <  Ls  scale = (float)Ls/Ld; running (i0, f0) = (0, 0); for each d: s1 = (d+1)*scale, i1 = (long)s1, f1 = s1 - i1;
                acc = (1-f0) src[i0], n = 1-f0; for si in (i0, i1): acc += src[si], n += 1; if i1 < Ls: acc += f1 src[i1], n += f1;
                out = acc / n; (i0, f0) = (i1, f1)
    all in fp32.  img: float32 [C, H, W].  Scalar loops, one rounding per operation: the checker of dataset.image_scale (numpy, vectorised
    over the other axes) and of cg_images_u8_scale_to_f32 (device)."""
    def axis(src, Ld):
        Ls = len(src)
        out = [f32(0)] * Ld
        if Ld == Ls:
            return list(src)
        if Ld > Ls:
            scale = f32(f32(Ls - 1) / f32(Ld - 1))
            for d in range(Ld - 1):
                sf = f32(f32(d) * scale)
                i = int(sf)
                f = f32(sf - f32(i))
                out[d] = src[0] if Ls == 1 else f32(f32(f32(f32(1) - f) * src[i]) + f32(f * src[i + 1]))
            out[Ld - 1] = src[Ls - 1]
            return out
        scale = f32(f32(Ls) / f32(Ld))
        i0, f0 = 0, f32(0)
        for d in range(Ld):
            s1 = f32(f32(d + 1) * scale)
            i1 = int(s1)
            f1 = f32(s1 - f32(i1))
            acc = f32(f32(f32(1) - f0) * src[i0])
            n = f32(f32(1) - f0)
            for si in range(i0 + 1, i1):
                acc = f32(acc + src[si]); n = f32(n + f32(1))
            if i1 < Ls:
                acc = f32(acc + f32(f1 * src[i1])); n = f32(n + f1)
            out[d] = f32(acc / n)
            i0, f0 = i1, f1
        return out
    img = np.asarray(img, dtype=f32)
    C, H, W = img.shape
    tmp = np.empty((C, H, w), f32)
    for c in range(C):
        for y in range(H):
            tmp[c, y] = axis([f32(v) for v in img[c, y]], w)
    out = np.empty((C, h, w), f32)
    for c in range(C):
        for x in range(w):
            out[c, :, x] = axis([f32(v) for v in tmp[c, :, x]], h)
    return out


def load_image(u8_hwc, w, h, color_space="rgb"):
    """One image of the pool: image.load(path, 3, 'float') (bytes / 255) -> image.scale -> rgbToColorSpace (nn_utils.lua:223-278; 'y':
    z = 0 + 0.21 r, + 0.72 g, + 0.07 b)."""
    img = (np.asarray(u8_hwc, dtype=np.uint8).astype(f32) / f32(255)).transpose(2, 0, 1)
    img = image_scale(img, w, h)
    if color_space == "y":
        z = f32(0.21) * img[0]
        z = z + f32(0.72) * img[1]
        z = z + f32(0.07) * img[2]
        return z[None].astype(f32)
    return img


# ---------------------------------------------------------------- criterion / optim
def bce_forward(p, t):
    """nn.BCECriterion, sizeAverage, eps=1e-12 [upstream] (train.lua:181)."""
    p, t = p.reshape(-1).astype(f32), t.reshape(-1).astype(f32)
    eps = f32(1e-12)
    return float(-np.sum(np.log(p + eps) * t + np.log(f32(1) - p + eps) * (f32(1) - t), dtype=np.float64) / p.size)


def bce_backward(p, t):
    shape = p.shape
    p, t = p.reshape(-1).astype(f32), t.reshape(-1).astype(f32)
    eps = f32(1e-12)
    return (-(t - p) / ((f32(1) - p + eps) * (p + eps)) / f32(p.size)).astype(f32).reshape(shape)


def adam(x, g, state, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8):
    """Torch7 optim.adam: m=b1*m+(1-b1)g; v=b2*v+(1-b2)g^2; x -= lr*sqrt(1-b2^t)/(1-b1^t) * m/(sqrt(v)+eps)."""
    if "t" not in state:
        state["t"] = 0
        state["m"] = np.zeros_like(x)
        state["v"] = np.zeros_like(x)
    state["t"] += 1
    t = state["t"]
    m, v = state["m"], state["v"]
    m *= f32(b1); m += f32(1 - b1) * g
    v *= f32(b2); v += f32(1 - b2) * g * g
    denom = np.sqrt(v) + f32(eps)
    step = f32(lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t))
    x -= step * m / denom


def sgd(x, g, state, lr=0.02, momentum=0.0):
    """Torch7 optim.sgd (train.lua:201-204): dampening defaults to momentum; first step v = g [upstream]."""
    if momentum != 0:
        if "v" not in state:
            state["v"] = g.copy()
        else:
            state["v"] *= f32(momentum); state["v"] += f32(1 - momentum) * g
        x -= f32(lr) * state["v"]
    else:
        x -= f32(lr) * g


def adagrad(x, g, state, lr=1e-3):
    """Torch7 optim.adagrad: var += g^2; x -= lr * g / (sqrt(var) + 1e-10) [upstream]."""
    if "var" not in state:
        state["var"] = np.zeros_like(x)
    state["var"] += g * g
    x -= f32(lr) * g / (np.sqrt(state["var"]) + f32(1e-10))


# --------------------------------------------------------------------- the hot path
class Trainer:
    """One iteration of adversarial.lua:51-275 with D_iterations=G_iterations=1, Adam, defaults of
    train.lua:26-45 (D_L1=0, D_L2=1e-4, G_L1=G_L2=0, D_clamp=1, G_clamp=5)."""

    def __init__(self, G, D, D_L1=0.0, D_L2=1e-4, G_L1=0.0, G_L2=0.0, D_clamp=1.0, G_clamp=5.0):
        self.G, self.D = G, D
        self.pG, self.gG = get_parameters(G)
        self.pD, self.gD = get_parameters(D)
        self.stG, self.stD = {}, {}
        self.o = dict(D_L1=D_L1, D_L2=D_L2, G_L1=G_L1, G_L2=G_L2, D_clamp=D_clamp, G_clamp=G_clamp)
        self.grad_allreduce = None  # data-parallel hook: f(gflat) -> mean over ranks, in place

    def feval_D(self, inputs, targets):
        """adversarial.lua:72-167."""
        o = self.o
        self.gD[...] = 0
        out = self.D.forward(inputs)
        f = bce_forward(out, targets)
        self.D.backward(bce_backward(out, targets.reshape(out.shape)))
        if self.grad_allreduce is not None:
            self.grad_allreduce(self.gD)
        if o["D_L1"] != 0 or o["D_L2"] != 0:
            f += o["D_L1"] * float(np.abs(self.pD).sum(dtype=np.float64))
            f += o["D_L2"] * float((self.pD.astype(np.float64) ** 2).sum()) / 2
            self.gD += (np.sign(self.pD) * f32(o["D_L1"]) + self.pD * f32(o["D_L2"])).astype(f32)
        if o["D_clamp"] != 0:
            np.clip(self.gD, -o["D_clamp"], o["D_clamp"], out=self.gD)
        return f, out

    def feval_G(self, noise, targets):
        """adversarial.lua:171-215."""
        o = self.o
        self.gG[...] = 0
        samples = self.G.forward(noise)
        out = self.D.forward(samples)
        f = bce_forward(out, targets)
        df_do = self.D.backward(bce_backward(out, targets.reshape(out.shape)))  # MODEL_D.modules[1].gradInput
        self.G.backward(df_do)
        if self.grad_allreduce is not None:
            self.grad_allreduce(self.gG)
        if o["G_L1"] != 0 or o["G_L2"] != 0:
            f += o["G_L1"] * float(np.abs(self.pG).sum(dtype=np.float64))
            f += o["G_L2"] * float((self.pG.astype(np.float64) ** 2).sum()) / 2
            # adversarial.lua:206 multiplies the sign term by G_L2 (latent upstream bug, inert at defaults)
            self.gG += (np.sign(self.pG) * f32(o["G_L2"]) + self.pG * f32(o["G_L2"])).astype(f32)
        if o["G_clamp"] != 0:
            np.clip(self.gG, -o["G_clamp"], o["G_clamp"], out=self.gG)
        return f, samples, out

    def step(self, real, noise_d, noise_g):
        """real [N/2,C,S,S] in [0,1]; noise_d [N/2,nd]; noise_g [N,nd].  Returns dict of observables."""
        half = real.shape[0]
        N = 2 * half
        # (1) D update: rows 1..N/2 real (target 1), rows N/2+1..N = G(noise) (target 0)  (:221-238)
        fake = self.G.forward(noise_d)  # train-mode BN on N/2, no gradient (nn_utils.lua:52)
        inputs = np.concatenate([real, fake], axis=0).astype(f32)
        targets = np.concatenate([np.ones(half, f32), np.zeros(half, f32)])
        fD, outD = self.feval_D(inputs, targets)
        gD = self.gD.copy()   # what optim.adam receives (penalty + clamp applied); feval_G's D:backward adds to gD later
        adam(self.pD, self.gD, self.stD)
        # (2) G update: fresh noise, targets all "real" (:253-266)
        fG, samples, outG = self.feval_G(noise_g, np.ones(N, f32))
        gG = self.gG.copy()
        adam(self.pG, self.gG, self.stG)
        return dict(fD=fD, fG=fG, outD=outD, outG=outG, fake=fake, samples=samples, gD=gD, gG=gG)
