"""utils/nn_utils.lua — what the training step uses: createNoiseInputs :35-39,
createImagesFromNoise :45-69, createImages :75-77, getNumberOfParameters :453-462, activateCuda :620-680 — and the
consumer side of the trained nets: sortImagesByPrediction :89-117, visualizeProgress :130-186 (without
the display server), toRgb :188-220, imagesToGridTensor / saveImagesAsGrid :526-583, rateWithV :686-711."""
import os
import struct
import zlib

import numpy as np

from . import nn
from .tensor import Tensor, lib, rng, stream


def createNoiseInputs(S, N):
    """U(-1,1) noise [N, noiseDim] (utils/nn_utils.lua:35-39), generated on the device from the engine's counter stream.  A FRESH
    tensor per call, as upstream: callers keep what they get (train.lua:220 holds VIS_NOISE_INPUTS for the whole run)."""
    return _fill_noise(Tensor.empty((N, S.OPT["noiseDim"])))


def stepNoiseInputs(S, N):
    """The training iteration's private form of createNoiseInputs: one persistent buffer per N (the D-step's half batch and the
    G-step's full batch never share one).  A captured iteration (cg_graph_*) must find its noise at the same address on every
    replay, and the step itself never keeps a noise tensor across two calls; nothing outside adversarial.iteration may use this."""
    cache = S.__dict__.setdefault("_noise_bufs", {})
    t = cache.get(N)
    if t is None:
        t = cache[N] = Tensor.empty((N, S.OPT["noiseDim"]))
    return _fill_noise(t)


def _fill_noise(t, offset=None):
    """offset: an explicit position of the counter stream (the caller reserves it and advances the generator itself)."""
    r = rng()
    lib().rng_uniform_dev(stream(), t.ptr, t.nElement(), -1.0, 1.0, r.seed, r.take(t.nElement()) if offset is None else int(offset), r.base_ptr())
    return t


def stepNoiseInputsAt(S, N, offset):
    """stepNoiseInputs with the draws taken from `offset` of the counter stream instead of its current position: the G-step's noise
    drawn ahead of time (adversarial.iteration runs both generator forwards side by side) at the position it has in the reference's
    order - behind the D-step's dropout masks."""
    cache = S.__dict__.setdefault("_noise_bufs", {})
    t = cache.get(N)
    if t is None:
        t = cache[N] = Tensor.empty((N, S.OPT["noiseDim"]))
    return _fill_noise(t, offset)


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


def switchToEvaluationMode(S):
    """nn_utils.lua:334-340: dropout off, batch-norm on running statistics."""
    S.MODEL_G.evaluate()
    S.MODEL_D.evaluate()


def switchToTrainingMode(S):
    """nn_utils.lua:343-349."""
    S.MODEL_G.training()
    S.MODEL_D.training()


def sortImagesByPrediction(S, images, ascending=False, nbMaxOut=None):
    """nn_utils.lua:89-117: rate images with D (in chunks of OPT.batchSize) and sort them by D's certainty that they
    are real; returns (images as a host array [n,C,H,W], predictions [n]).  Used by sample.lua:104-105 and the
    epoch visualisation (nn_utils.lua:138-147)."""
    imgs = nn.as_nhwc(nn.to_device(images))
    N, bs = imgs.shape[0], S.OPT["batchSize"]
    preds = []
    for a in range(1, N + 1, bs):
        b = min(a + bs - 1, N)
        preds.append(nn.as_plain(S.MODEL_D.forward(imgs.rows(a, b))).numpy().reshape(-1))
    preds = np.concatenate(preds)
    order = np.argsort(preds, kind="stable")
    if not ascending:
        order = order[::-1]
    if nbMaxOut is not None:
        order = order[:nbMaxOut]
    return imgs.numpy()[order], preds[order]


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


# ------------------------------------------------------------------ visual grids
def toRgb(images, colorSpace):
    """nn_utils.lua:188-220 for the colour spaces the engine loads ('rgb' as is, 'y' repeated over three channels)."""
    images = np.asarray(images, dtype=np.float32)
    if images.ndim == 3:
        images = images[None]
    if colorSpace == "rgb":
        return images
    if colorSpace == "y":
        return np.tile(images, (1, 3, 1, 1))
    raise NotImplementedError(f"colour space '{colorSpace}' is outside the hot-path scope (rgb | y)")


# the 3 x 5 digit glyphs the reference draws the epoch number with (CHAR_TENSORS, nn_utils.lua:465-515), one 15-bit row-major
# mask per digit
_DIGITS = (0b111101101101111, 0b001001001001001, 0b111001111100111, 0b111001011001111, 0b101101111001001,
           0b111100111001111, 0b111100111101111, 0b111001001001001, 0b111101111101111, 0b111101111001111)


def _glyph(d):
    bits = _DIGITS[d]
    return np.array([[(bits >> (14 - (r * 3 + c))) & 1 for c in range(3)] for r in range(5)], dtype=np.float32)


def imagesToGridTensor(images, height, width, epoch, dims=None):
    """nn_utils.lua:526-569: the first height*width images row by row on a black canvas of height*H + 7 rows, the epoch
    number in 3 x 5 digits at the bottom right (last digit rightmost).  images [n,C,H,W] in [0,1]; returns [C,Hpx,Wpx]."""
    images = np.asarray(images, dtype=np.float32)
    C = images.shape[1]
    H, W = (dims[1], dims[2]) if dims is not None else images.shape[2:]
    Hpx, Wpx = height * H + (1 + 5 + 1), width * W
    grid = np.zeros((C, Hpx, Wpx), np.float32)
    for i in range(min(images.shape[0], height * width)):
        y, x = divmod(i, width)
        grid[:, y * H:(y + 1) * H, x * W:(x + 1) * W] = images[i]
    for pos, ch in enumerate(reversed(str(int(epoch))), start=1):
        y0 = Hpx - 1 - 5 - 1                 # 0-based form of yStart = heightPx - 1 - 5
        x0 = Wpx - 1 - pos * 5 - pos - 1     # ... and of xStart = widthPx - 1 - pos*5 - pos
        if x0 < 0:
            break
        grid[:, y0:y0 + 5, x0:x0 + 3] = _glyph(int(ch))
    return grid


def _png_bytes(img):
    """[C,H,W] floats in [0,1] (C = 1 or 3) as an 8-bit PNG (what image.save writes): clamp, x255, round to nearest."""
    C, H, W = img.shape
    assert C in (1, 3)
    px = np.clip(np.rint(np.clip(img, 0.0, 1.0) * 255.0), 0, 255).astype(np.uint8).transpose(1, 2, 0)
    raw = b"".join(b"\x00" + px[y].tobytes() for y in range(H))   # filter type 0 per scan line

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    ihdr = struct.pack(">IIBBBBB", W, H, 8, 2 if C == 3 else 0, 0, 0, 0)
    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", ihdr) + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b"")


def saveImagesAsGrid(filepath, images, height, width, epoch, dims=None):
    """nn_utils.lua:579-583."""
    grid = imagesToGridTensor(images, height, width, epoch, dims)
    d = os.path.dirname(filepath)
    if d:
        os.makedirs(d, exist_ok=True)
    with open(filepath, "wb") as f:
        f.write(_png_bytes(grid))
    return grid


def rateWithV(S, images):
    """nn_utils.lua:686-711: 1 - mean of V's first output (1 = fake) over the images; V is any module with forward()."""
    V = getattr(S, "MODEL_V", None)
    if V is None:
        return None
    imgs = nn.to_device(np.asarray(images, dtype=np.float32))
    N, bs = imgs.shape[0], S.OPT["batchSize"]
    sm = 0.0
    for a in range(1, N + 1, bs):
        b = min(a + bs - 1, N)
        p = nn.as_plain(V.forward(imgs.rows(a, b))).numpy().reshape(b - a + 1, -1)
        sm += float(p[:, 0].sum())
    return 1.0 - sm / N


def visualizeProgress(S, noiseInputs, trainImages, save_dir, start_time=0, plot_data=None, verbose=False):
    """nn_utils.lua:130-186 without the display server: evaluation mode, the fixed-noise images, the best / worst 50 of them
    according to D (with one real image and one synthetic non-image planted as sanity checks, :133-146), the three PNG
    grids (10x10, 7x7, 7x7) under save_dir/images{,_good,_bad}/<start>_<epoch>.png, V's ratings when S.MODEL_V is set."""
    switchToEvaluationMode(S)
    try:
        train = np.asarray(trainImages, dtype=np.float32)[:50]
        C, H, W = train.shape[1:]
        sanity = np.random.RandomState(S.EPOCH).uniform(0.0, 0.5, size=(C, H, W)).astype(np.float32)   # not S.random: it draws the batches
        for i in range(1, H + 1):               # the reference walks OPT.scale (= the image side) in both directions
            for j in range(1, W + 1):
                if i == j:
                    sanity[0, i - 1, j - 1] = 1.0
                elif i % 4 == 0 and j % 4 == 0:
                    sanity[0, i - 1, j - 1] = 0.5
        rnd = nn.as_nhwc(createImagesFromNoise(S, noiseInputs)).numpy()
        clone = rnd.copy()
        clone[-2] = train[0]
        clone[-1] = sanity
        good, _ = sortImagesByPrediction(S, clone, False, 50)
        bad, _ = sortImagesByPrediction(S, clone, True, 50)
        if np.isnan(rnd).any():
            print("[nn_utils vizProgress] Generated images contain NaNs")
        cs, ep = S.OPT["colorSpace"], S.EPOCH
        out = {}
        for sub, imgs, side in (("images", rnd, 10), ("images_good", good, 7), ("images_bad", bad, 7)):
            path = os.path.join(save_dir, sub, "%d_%05d.png" % (start_time, ep))
            saveImagesAsGrid(path, toRgb(imgs, cs), side, side, ep)
            out[sub] = path
        ratings = [rateWithV(S, x) for x in (rnd, good, bad)]
        if ratings[0] is not None:
            if plot_data is not None:
                plot_data.append([ep] + ratings)
            if verbose:
                print("<nnutils viz> [V] semiRandom: %.4f, goodImages: %.4f, badImages: %.4f" % tuple(ratings))
        out["ratings"] = ratings
        return out
    finally:
        switchToTrainingMode(S)


# ------------------------------------------------------------------ sample.lua's helpers
def toDisplayTensor(images, nrow, padding=0):
    """image.toDisplayTensor{input=images, nrow=nrow} as sample.lua:166-168 uses it: `nrow` images PER ROW on a black canvas,
    the whole grid rescaled to [0,1] by its global minimum / maximum (the package's default min-max normalisation; recalled
    upstream behaviour - the image rock is not vendored).  images [n,C,H,W]; returns [C, rows*H, nrow*W]."""
    images = np.asarray(images, dtype=np.float32)
    n, C, H, W = images.shape
    rows = -(-n // nrow)
    grid = np.zeros((C, rows * (H + padding), nrow * (W + padding)), np.float32)
    for i in range(n):
        y, x = divmod(i, nrow)
        grid[:, y * (H + padding):y * (H + padding) + H, x * (W + padding):x * (W + padding) + W] = images[i]
    lo, hi = float(grid.min()), float(grid.max())
    return (grid - lo) / (hi - lo) if hi > lo else np.zeros_like(grid)


def selectRandomImagesFrom(images, n, rs):
    """sample.lua:199-207: the first n entries of a random permutation (torch.randperm -> the given generator)."""
    images = np.asarray(images)
    shuffle = rs.permutation(images.shape[0])
    return images[shuffle[:min(n, images.shape[0])]]


def findClosestNeighboursOf(images, trainingSet):
    """sample.lua:131-151: for every image its nearest training image in the 2-norm; [(image, neighbour, distance)]."""
    train = np.asarray(trainingSet, dtype=np.float32)
    flat = train.reshape(train.shape[0], -1)
    out = []
    for img in np.asarray(images, dtype=np.float32):
        d = np.sqrt(((flat - img.reshape(1, -1)) ** 2).sum(axis=1))
        j = int(np.argmin(d))
        out.append((img, train[j].copy(), float(d[j])))
    return out
