"""nn modules <-> Torch7 objects: the field tables torch.save writes for the classes models.lua builds
(nn / cudnn / stn, era 2015-16), so that a checkpoint exported here has the shape of the reference's
{D, G, opt, plot_data, epoch} file (train.lua:252-261) and a file of that shape can be brought in.

Export mirrors NN_UTILS.prepareNetworkForSave (nn_utils.lua:428-451): output / gradInput / work buffers are written empty;
parameters in Torch7's canonical layouts ([out][in][kH][kW], [out][in]); gradients zero-sized are avoided (same sizes as the
parameters, zeros) because Module:getParameters() flattens them on load.  `cuda=True` writes the tensors as torch.CudaTensor,
which is how the reference's GPU runs save them (loading that needs cutorch)."""
import numpy as np

from . import cudnn, nn, t7
from .t7 import Storage, TorchObject, array_table, table_list


def _empty(cuda):
    a = np.zeros((0,), np.float32)
    return t7.CudaTensor(a) if cuda else a


def _t(array, cuda):
    a = np.ascontiguousarray(array, dtype=np.float32)
    return t7.CudaTensor(a) if cuda else a


def _arr(v):
    return v.array if isinstance(v, t7.CudaTensor) else np.asarray(v)


def to_t7(m, cuda=False):
    """One module (recursively) as a TorchObject."""
    f = {"output": _empty(cuda), "gradInput": _empty(cuda), "train": bool(m.train), "_type": "torch.CudaTensor" if cuda else "torch.FloatTensor"}
    name = m.typename
    T = lambda x: _t(x.numpy(), cuda)
    Z = lambda x: _t(np.zeros(x.shape, np.float32), cuda)
    if isinstance(m, nn.Sequential):        # Sequential, ConcatTable, Concat
        f["modules"] = array_table([to_t7(c, cuda) for c in m.modules])
        if isinstance(m, nn.Concat):
            f["dimension"] = m.dimension
            f["size"] = Storage(np.zeros((0,), np.int64))
        if isinstance(m, nn.ConcatTable):
            f["output"], f["gradInput"] = {}, _empty(cuda)
    elif isinstance(m, nn.SpatialConvolution):
        f.update(nInputPlane=m.nInputPlane, nOutputPlane=m.nOutputPlane, kW=m.kW, kH=m.kH, dW=m.dW, dH=m.dH, padW=m.padW, padH=m.padH,
                 weight=T(m.weight), bias=T(m.bias), gradWeight=Z(m.weight), gradBias=Z(m.bias), finput=_empty(cuda), fgradInput=_empty(cuda))
        if isinstance(m, nn.SpatialConvolutionUpsample):
            f.update(factor=m.factor, nInputPlaneU=m.nInputPlaneU, nOutputPlaneU=m.nOutputPlaneU)
        if name.startswith("cudnn."):
            f["groups"] = 1
    elif isinstance(m, nn.Linear):
        f.update(weight=T(m.weight), bias=T(m.bias), gradWeight=Z(m.weight), gradBias=Z(m.bias))
    elif isinstance(m, nn.PReLU):
        f.update(nOutputPlane=0, weight=T(m.weight), gradWeight=Z(m.weight), gradWeightBuf=_empty(cuda), gradWeightBuf2=_empty(cuda))
    elif isinstance(m, nn.LeakyReLU):
        f.update(negative_scale=float(m.negative_scale), negative=_empty(cuda))
    elif isinstance(m, nn.SpatialBatchNormalization):
        f.update(affine=True, eps=float(m.eps), momentum=float(m.momentum), weight=T(m.weight), bias=T(m.bias),
                 gradWeight=Z(m.weight), gradBias=Z(m.bias), running_mean=T(m.running_mean), running_var=T(m.running_var))
    elif isinstance(m, nn.View):
        f.update(size=Storage(np.asarray(m.sizes, np.int64)), numElements=int(np.prod(m.sizes)))
    elif isinstance(m, nn.Copy):
        f.update(intype=m.intype, outtype=m.outtype, dontCast=True)
        f["output"] = _empty(m.outtype == "torch.CudaTensor" and cuda)
        f["gradInput"] = _empty(m.intype == "torch.CudaTensor" and cuda)
    elif isinstance(m, nn.Transpose):
        f["permutations"] = array_table([array_table(list(p)) for p in m.permutations])
    elif isinstance(m, nn.SpatialUpSamplingNearest):
        f.update(scale_factor=m.scale_factor, inputSize=Storage(np.zeros(4, np.int64)), outputSize=Storage(np.zeros(4, np.int64)))
    elif isinstance(m, (nn.SpatialAveragePooling, nn.SpatialMaxPooling)):
        f.update(kW=2, kH=2, dW=2, dH=2, padW=0, padH=0, ceil_mode=False)
        if isinstance(m, nn.SpatialAveragePooling):
            f.update(count_include_pad=True, divide=True)
        else:
            f["indices"] = _empty(cuda)
    elif isinstance(m, nn.SpatialDropout):
        f.update(p=float(m.p), noise=_empty(cuda))
    elif isinstance(m, nn.Dropout):
        f.update(p=float(m.p), noise=_empty(cuda), v2=True)
    elif isinstance(m, nn.AffineTransformMatrixGenerator):
        f.update(useRotation=m.useRotation, useScale=m.useScale, useTranslation=m.useTranslation)
    elif isinstance(m, nn.AffineGridGeneratorBHWD):
        h, w = m.height, m.width
        base = np.empty((h, w, 3), np.float32)        # stn's baseGrid: (y, x, 1) with y, x in [-1, 1]
        base[:, :, 0] = (-1 + np.arange(h, dtype=np.float32) / max(h - 1, 1) * 2)[:, None]
        base[:, :, 1] = (-1 + np.arange(w, dtype=np.float32) / max(w - 1, 1) * 2)[None, :]
        base[:, :, 2] = 1
        f.update(height=h, width=w, baseGrid=_t(base, cuda), batchGrid=_empty(cuda))
    elif isinstance(m, (nn.Sigmoid, nn.BilinearSamplerBHWD)):
        pass
    else:
        raise TypeError(f"no Torch7 form for {type(m).__name__}")
    return TorchObject(name, f)


def _num(o, k, default=None):
    v = o.get(k, default)
    return int(v) if isinstance(v, (int, float)) and float(v).is_integer() else v


def from_t7(o):
    """A TorchObject tree (this module's export, or a torch.save of the same classes) as engine modules."""
    n = o.typename
    if n in ("nn.Sequential", "nn.ConcatTable", "nn.Concat"):
        m = nn.Concat(_num(o, "dimension")) if n == "nn.Concat" else (nn.ConcatTable() if n == "nn.ConcatTable" else nn.Sequential())
        for c in table_list(o["modules"]):
            m.add(from_t7(c))
    elif n in ("nn.SpatialConvolution", "cudnn.SpatialConvolution", "nn.SpatialConvolutionMM"):
        cls = cudnn.SpatialConvolution if n.startswith("cudnn.") else nn.SpatialConvolution
        m = cls(_num(o, "nInputPlane"), _num(o, "nOutputPlane"), _num(o, "kW"), _num(o, "kH"), _num(o, "dW", 1), _num(o, "dH", 1),
                _num(o, "padW", 0), _num(o, "padH", _num(o, "padW", 0)))
        m.weight.copy(_arr(o["weight"]).reshape(m.weight.shape))     # SpatialConvolutionMM keeps [out][in*kH*kW]: same memory
        m.bias.copy(_arr(o["bias"]))
    elif n in ("nn.SpatialConvolutionUpsample", "cudnn.SpatialConvolutionUpsample"):
        m = nn.SpatialConvolutionUpsample(_num(o, "nInputPlaneU"), _num(o, "nOutputPlaneU"), _num(o, "kW"), _num(o, "kH"), _num(o, "factor", 2))
        m.weight.copy(_arr(o["weight"]).reshape(m.weight.shape)); m.bias.copy(_arr(o["bias"]))
    elif n == "nn.Linear":
        w = _arr(o["weight"])
        m = nn.Linear(w.shape[1], w.shape[0])
        m.weight.copy(w); m.bias.copy(_arr(o["bias"]))
    elif n == "nn.PReLU":
        m = nn.PReLU()
        m.weight.copy(_arr(o["weight"]).reshape(1))
    elif n == "nn.LeakyReLU":
        m = nn.LeakyReLU(o.get("negative_scale", o.get("negval", 0.333)))
    elif n == "nn.Sigmoid":
        m = nn.Sigmoid()
    elif n == "nn.SpatialBatchNormalization":
        rm = _arr(o["running_mean"])
        m = nn.SpatialBatchNormalization(rm.size, o.get("eps", 1e-5), o.get("momentum", 0.1))
        m.weight.copy(_arr(o["weight"])); m.bias.copy(_arr(o["bias"])); m.running_mean.copy(rm)
        if o.get("running_var") is not None:
            m.running_var.copy(_arr(o["running_var"]))
        else:                                       # older nn: running_std = 1 / sqrt(var + eps)
            rs = _arr(o["running_std"]).astype(np.float64)
            m.running_var.copy((1.0 / (rs * rs) - m.eps).astype(np.float32))
    elif n == "nn.View":
        m = nn.View(*[int(s) for s in np.asarray(o["size"]).reshape(-1)])
    elif n == "nn.Copy":
        kind = lambda t: "torch.CudaTensor" if isinstance(t, str) and "Cuda" in t else "torch.FloatTensor"
        m = nn.Copy(kind(o.get("intype")), kind(o.get("outtype")), True, True)
    elif n == "nn.Transpose":
        m = nn.Transpose(*[tuple(table_list(p)) for p in table_list(o["permutations"])])
    elif n == "nn.SpatialUpSamplingNearest":
        m = nn.SpatialUpSamplingNearest(_num(o, "scale_factor"))
    elif n == "nn.SpatialAveragePooling":
        m = nn.SpatialAveragePooling(_num(o, "kW"), _num(o, "kH"), _num(o, "dW"), _num(o, "dH"))
    elif n in ("nn.SpatialMaxPooling", "cudnn.SpatialMaxPooling"):
        m = nn.SpatialMaxPooling(_num(o, "kW"), _num(o, "kH"), _num(o, "dW"), _num(o, "dH"))
    elif n == "nn.SpatialDropout":
        m = nn.SpatialDropout(o.get("p", 0.5))
    elif n == "nn.Dropout":
        m = nn.Dropout(o.get("p", 0.5))
    elif n == "nn.AffineTransformMatrixGenerator":
        m = nn.AffineTransformMatrixGenerator(bool(o.get("useRotation")), bool(o.get("useScale")), bool(o.get("useTranslation")))
    elif n == "nn.AffineGridGeneratorBHWD":
        m = nn.AffineGridGeneratorBHWD(_num(o, "height"), _num(o, "width"))
    elif n == "nn.BilinearSamplerBHWD":
        m = nn.BilinearSamplerBHWD()
    else:
        raise TypeError(f"no engine module for Torch7 class {n}")
    if o.get("train") is False:
        m.evaluate()
    return m
