"""utils/nn_utils.lua — the subset the training step uses:
createNoiseInputs :35-39, createImagesFromNoise :45-69, createImages :75-77, getNumberOfParameters :453-462,
activateCuda :620-680.  Visualisation / checkpoint helpers are outside the hot path (row 5b)."""
from . import nn
from .tensor import Tensor, lib, rng, stream


def createNoiseInputs(S, N):
    """U(-1,1) noise [N, noiseDim], generated on the device from the engine's counter stream."""
    t = Tensor.empty((N, S.OPT["noiseDim"]))
    r = rng()
    lib().rng_uniform_dev(stream(), t.ptr, t.nElement(), -1.0, 1.0, r.seed, r.take(t.nElement()), r.base_ptr())
    return t


def createImagesFromNoise(S, noiseInputs, outputAsList=False, *_):
    """G forward in chunks of OPT.batchSize (nn_utils.lua:48-58).  Note MODEL_G stays in training mode
    (train-mode BN on the chunk), exactly as upstream.  The reference's :clone() of each chunk is kept only when
    there is more than one chunk (the G output buffer is reused by the next forward)."""
    noiseInputs = nn.to_device(noiseInputs)
    N = noiseInputs.size(1)
    bs = S.OPT["batchSize"]
    nBatches = -(-N // bs)
    if nBatches == 1:
        images = S.MODEL_G.forward(noiseInputs)
    else:
        images = None
        for i in range(1, nBatches + 1):
            a, b = 1 + (i - 1) * bs, min(i * bs, N)
            generated = nn.as_nhwc(S.MODEL_G.forward(noiseInputs.rows(a, b)))
            if images is None:
                images = Tensor.empty((N,) + generated.shape[1:], "nhwc")
            images.rows(a, b).copy(generated)
    if outputAsList:
        arr = images.numpy()
        return [arr[i] for i in range(arr.shape[0])]
    return images


def createImages(S, N, outputAsList=False):
    return createImagesFromNoise(S, createNoiseInputs(S, N), outputAsList)


def getNumberOfParameters(net):
    """nn_utils.lua:453-462: sums weight and bias sizes over listModules()."""
    n = 0
    for m in net.listModules():
        for name in ("weight", "bias"):
            t = getattr(m, name, None)
            if t is not None:
                n += t.nElement()
    return n


def activateCuda(net):
    """nn_utils.lua:620-680: wrap the net in Copy layers unless it already contains some."""
    if any(isinstance(m, nn.Copy) for m in net.listModules()):
        return net
    tmp = nn.Sequential()
    tmp.add(nn.Copy("torch.FloatTensor", "torch.CudaTensor"))
    tmp.add(net)
    tmp.add(nn.Copy("torch.CudaTensor", "torch.FloatTensor"))
    return tmp
