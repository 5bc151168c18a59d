"""dataset.lua restated: list + sort the image files of the configured directories
(dataset.lua:57-83), load `count` random ones (torch.randperm, :158-168), scale to width x height (image.scale,
bilinear, :129-131) and convert the colour space (NN_UTILS.rgbToColorSpace, nn_utils.lua:223-278: 'rgb' or 'y' with
weights 0.21/0.72/0.07).  Host-side input pipeline: decoding uses PIL; the result is a float array in [0,1] that
adversarial.TrainData uploads to HBM once per epoch (the reference reloads N_epoch images per epoch, train.lua:225)."""
import os

import numpy as np

dirs = []
fileExtension = ""
height = 32
width = 32
colorSpace = "rgb"
paths = None
_rs = np.random.RandomState(1)


def setDirs(d):
    global dirs, paths
    dirs, paths = list(d), None


def setFileExtension(ext):
    global fileExtension
    fileExtension = ext


def setHeight(h):
    global height
    height = int(h)


def setWidth(w):
    global width
    width = int(w)


def seed(s):
    global _rs
    _rs = np.random.RandomState(s)


def loadPaths():
    """dataset.lua:57-83: every file with the extension in every directory, sorted."""
    global paths
    files = []
    for d in dirs:
        for f in os.listdir(d):
            if f.lower().endswith("." + fileExtension.lower()):
                files.append(os.path.join(d, f))
    files.sort()
    paths = files
    return paths


def rgb2y(im):
    """nn_utils.lua:253-277."""
    return (np.float32(0.21) * im[0] + np.float32(0.72) * im[1] + np.float32(0.07) * im[2])[None].astype(np.float32)


def rgbToColorSpace(images, cs):
    """nn_utils.lua:223-249 ('hsl'/'yuv' are not reachable from the benchmarked configurations)."""
    if cs == "rgb":
        return images
    if cs == "y":
        return np.stack([rgb2y(im) for im in images])
    raise NotImplementedError(f"colour space '{cs}' is outside the hot-path scope (rgb | y)")


class _Data:
    def __init__(self, scaled):
        self.scaled = scaled

    def size(self):
        return self.scaled.shape[0]

    def __len__(self):
        return self.scaled.shape[0]

    def __getitem__(self, i):
        return self.scaled[i]


def loadRandomImages(count):
    """dataset.lua:123-170."""
    from PIL import Image
    if paths is None:
        loadPaths()
    if not paths:
        raise FileNotFoundError(f"no *.{fileExtension} images under {dirs}")
    shuffle = _rs.permutation(len(paths))
    data = np.empty((min(len(paths), count), 3, height, width), np.float32)
    for i in range(data.shape[0]):
        im = Image.open(paths[shuffle[i]]).convert("RGB").resize((width, height), Image.BILINEAR)
        data[i] = np.asarray(im, dtype=np.float32).transpose(2, 0, 1) / np.float32(255.0)
    return _Data(rgbToColorSpace(data, colorSpace))
