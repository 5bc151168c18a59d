"""dataset.lua restated: list + sort the image files of the configured directories
(dataset.lua:57-83), load `count` random ones (torch.randperm, :158-168), image.load them as floats in [0,1] (:166), scale the
FLOAT image to width x height (image.scale, :129-131) and convert the colour space (NN_UTILS.rgbToColorSpace,
nn_utils.lua:223-278: 'rgb' or 'y' with weights 0.21/0.72/0.07).  PIL only decodes; the scaling is `image_scale` below - the `image`
rock's default separable 'bilinear' [upstream, recalled: box averaging when shrinking, linear interpolation when enlarging] in fp32,
the arithmetic cg_images_u8_scale_to_f32 performs on the device (oracle/oracle.py holds the independent restatement the tests
compare both with).  The result is a float array in [0,1] that adversarial.TrainData uploads to HBM once per epoch (the reference
reloads N_epoch images per epoch, train.lua:225).

AsyncLoader is the production form of the same thing: the NEXT epoch's images are decoded by a worker
thread straight into a page-locked buffer as 8-bit RGB at their own size (a quarter of the fp32 bytes over PCIe), copied on a copy
stream while the current epoch trains, scaled and converted on the device (cg_images_u8_scale_to_f32: /255, image.scale, colour
space) into the second of two HBM pools, and handed to the compute stream through an event - no host synchronisation on the
training path."""
import os
import threading

import numpy as np

dirs = []
fileExtension = ""
height = 32
width = 32
colorSpace = "rgb"
paths = None
_rs = np.random.RandomState(1)
_prefetch_state = None   # AsyncLoader: the generator's state BEFORE the pick of the prefetch that has not been consumed yet


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
    global _rs, _prefetch_state
    _rs = np.random.RandomState(s)
    _prefetch_state = None


def checkpoint_state():
    """What a checkpoint must store so that a resumed run loads the pools an uninterrupted one would: AsyncLoader has already
    drawn the NEXT epoch's permutation when an epoch ends, and a resumed loader draws again on construction - so the state to
    restore is the one from before that not-yet-consumed pick (the blocking loader has no pick in flight)."""
    return _prefetch_state if _prefetch_state is not None else _rs.get_state()


def restore_state(state):
    global _prefetch_state
    _rs.set_state(state)
    _prefetch_state = None


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


def _pick(count):
    """The files of one load: a fresh permutation of the sorted paths, its first `count` entries (dataset.lua:158-163)."""
    if paths is None:
        loadPaths()
    if not paths:
        raise FileNotFoundError(f"no *.{fileExtension} images under {dirs}")
    shuffle = _rs.permutation(len(paths))
    return [paths[shuffle[i]] for i in range(min(len(paths), count))]


def _prefetch_pick(count):
    """_pick for a load that is consumed one epoch later (AsyncLoader): remembers the generator's state from before the pick,
    which is what a checkpoint taken in between must restore (checkpoint_state)."""
    global _prefetch_state
    _prefetch_state = _rs.get_state()
    return _pick(count)


def _decode(path):
    """One file as 8-bit RGB [H, W, 3] at its own size (what image.load reads before it divides by 255)."""
    from PIL import Image
    return np.asarray(Image.open(path).convert("RGB"), dtype=np.uint8)


def _scale_axis(a, Ld):
    """image.scale along the LAST axis of a float32 array [upstream, recalled: generic/image.c scaleLinear_rowcol]: every operation a
    single fp32 one in the loop's order (the device kernel images_u8_scale_k does the same sequence)."""
    f32 = np.float32
    Ls = a.shape[-1]
    if Ld == Ls:
        return a.copy()
    out = np.empty(a.shape[:-1] + (Ld,), f32)
    if Ld > Ls:      # linear interpolation, corner aligned; the last sample is copied
        scale = f32(Ls - 1) / f32(Ld - 1) if Ld > 1 else f32(0)
        for d in range(Ld - 1):
            sf = f32(d) * scale
            si = int(sf)
            sf = f32(sf - f32(si))
            out[..., d] = a[..., 0] if Ls == 1 else (f32(1) - sf) * a[..., si] + sf * a[..., si + 1]
        out[..., Ld - 1] = a[..., Ls - 1]
        return out
    scale = f32(Ls) / f32(Ld)      # box average, fractional ends weighted
    i0, f0 = 0, f32(0)
    for d in range(Ld):
        f1 = f32(d + 1) * scale
        i1 = int(f1)
        f1 = f32(f1 - f32(i1))
        acc = (f32(1) - f0) * a[..., i0]
        n = f32(1) - f0
        for si in range(i0 + 1, i1):
            acc = acc + a[..., si]
            n = f32(n + f32(1))
        if i1 < Ls:
            acc = acc + f1 * a[..., i1]
            n = f32(n + f1)
        out[..., d] = acc / n
        i0, f0 = i1, f1
    return out


def image_scale(img, w, h):
    """image.scale(img, w, h) for a float32 [C, H, W] image: rows to the target width first, then columns to the target height."""
    t = _scale_axis(np.ascontiguousarray(img, dtype=np.float32), w)
    return np.ascontiguousarray(_scale_axis(np.ascontiguousarray(t.transpose(0, 2, 1)), h).transpose(0, 2, 1))


def loadRandomImages(count):
    """dataset.lua:123-170."""
    files = _pick(count)
    data = np.empty((len(files), 3, height, width), np.float32)
    for i, f in enumerate(files):
        img = _decode(f).astype(np.float32).transpose(2, 0, 1) / np.float32(255.0)      # image.load(path, 3, 'float')
        data[i] = image_scale(img, width, height)
    return _Data(rgbToColorSpace(data, colorSpace))


class AsyncLoader:
    """Double-buffered epoch pools in HBM.  next() returns the pool (an engine tensor [n,C,H,W], NHWC memory) whose upload
    was started one call earlier and immediately starts on the following one: decode (worker thread) -> pinned u8 buffer ->
    cg_memcpy_h2d on the copy stream -> cg_images_u8_to_f32 -> event.  The image choice and the pixel values are exactly
    loadRandomImages' (same generator, same decode, same fp32 operations), so switching loaders does not change a run."""

    def __init__(self, count, depth=2):
        import ctypes
        from .tensor import Tensor, lib
        self._ct, self._T, self.L = ctypes, Tensor, lib()
        self.count, self.depth = int(count), int(depth)
        self.C = 1 if colorSpace == "y" else 3
        if paths is None:
            loadPaths()
        if not paths:
            raise FileNotFoundError(f"no *.{fileExtension} images under {dirs}")
        self.Hs, self.Ws = _decode(paths[0]).shape[:2]      # the dataset is written at one size (generate_dataset.py: 64 x 64)
        # The device kernel scales a batch of ONE source size, and at most 6x down (images_u8_scale_k's box loop).  The reference
        # (dataset.lua:129-131) scales every image on its own: an image of another size - or all of them, if the source is more than
        # 6x the target - takes the blocking loader's host path (image_scale, same arithmetic) and is patched into the pool.
        self.host_all = self.Hs > 6 * height or self.Ws > 6 * width
        if self.host_all:
            import warnings
            warnings.warn(f"AsyncLoader: {self.Ws}x{self.Hs} sources are more than 6x the {width}x{height} target: scaling on the host")
        # host_all: nothing is scaled on the device, so no 8-bit staging / device buffer of the (large) SOURCE size exists at all - the
        # worker writes the finished fp32 NHWC rows into a pinned buffer of the POOL's size and one asynchronous copy moves them
        # (the u8 path would pin count x Hs x Ws x 3 bytes per slot for images it never uploads).  Otherwise: the u8
        # staging, plus a pinned fp32 side buffer of the pool's size for the odd-sized images that take the host path.
        self.row = height * width * self.C * 4
        self.nbytes = self.count * self.row if self.host_all else self.count * self.Hs * self.Ws * 3
        self.copy_stream = ctypes.c_void_p()
        self.L.stream_create(ctypes.byref(self.copy_stream))
        self.slots = []
        f32p = ctypes.POINTER(ctypes.c_float)
        for _ in range(self.depth):
            host, dev_u8, ev_ready, ev_free, side = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
            self.L.host_alloc(ctypes.byref(host), self.nbytes)
            self.L.event_create(ctypes.byref(ev_ready)); self.L.event_create(ctypes.byref(ev_free))
            pool = Tensor.empty((self.count, self.C, height, width), "nhwc")
            if self.host_all:
                staging = np.ctypeslib.as_array(ctypes.cast(host, f32p), shape=(self.count, height, width, self.C))
                patch = None
            else:
                self.L.malloc(ctypes.byref(dev_u8), self.nbytes)
                staging = np.ctypeslib.as_array(ctypes.cast(host, ctypes.POINTER(ctypes.c_uint8)), shape=(self.count, self.Hs, self.Ws, 3))
                self.L.host_alloc(ctypes.byref(side), self.count * self.row)
                patch = np.ctypeslib.as_array(ctypes.cast(side, f32p), shape=(self.count, height, width, self.C))
            self.slots.append(dict(host=host, staging=staging, dev_u8=dev_u8, side=side, patch=patch, ready=ev_ready, free=ev_free, pool=pool,
                                   n=0, used=False))
        self.k = 0
        self._thread = None
        self._err = None
        self._start(self.slots[0])

    def _start(self, slot):
        files = _prefetch_pick(self.count)   # on the caller's thread: the generator's draws stay in program order
        slot["n"] = len(files)
        if slot["used"]:
            self.L.event_sync(slot["ready"])   # its previous upload has left the pinned buffers (long ago; costs nothing)

        slot["patches"] = []

        def work():
            try:
                for i, f in enumerate(files):
                    im = _decode(f)
                    if self.host_all or im.shape[:2] != (self.Hs, self.Ws):
                        # loadRandomImages' arithmetic for this one image: /255 -> image.scale -> colour space, as NHWC rows
                        img = image_scale(im.astype(np.float32).transpose(2, 0, 1) / np.float32(255.0), width, height)
                        img = rgbToColorSpace(img[None], colorSpace)[0].transpose(1, 2, 0)
                        if self.host_all:
                            slot["staging"][i] = img
                        else:
                            slot["patch"][i] = img         # pinned: the copy below needs no synchronisation
                            slot["patches"].append(i)
                            slot["staging"][i] = 0
                    else:
                        slot["staging"][i] = im
            except Exception as e:   # surfaced by next()
                self._err = e
        self._thread = threading.Thread(target=work, daemon=True)
        self._thread.start()

    def _upload(self, slot):
        L, cs, n = self.L, self.copy_stream, slot["n"]
        if slot["used"]:
            L.stream_wait_event(cs, slot["free"])      # the training stream has finished reading this pool
        if self.host_all:
            L.memcpy_h2d(cs, slot["pool"].ptr, slot["host"], n * self.row)
        else:
            L.memcpy_h2d(cs, slot["dev_u8"], slot["host"], n * self.Hs * self.Ws * 3)
            L.images_u8_scale_to_f32(cs, slot["dev_u8"], slot["pool"].ptr, n, self.Hs, self.Ws, height, width, 1 if colorSpace == "y" else 0)
            for i in slot["patches"]:     # images scaled on the host (another source size): over the rows the kernel produced from zeros
                L.memcpy_h2d(cs, slot["pool"].ptr + i * self.row, slot["side"].value + i * self.row, self.row)
            slot["patches"] = []
        L.event_record(slot["ready"], cs)

    def next(self):
        from .tensor import stream
        slot = self.slots[self.k]
        self._thread.join()
        if self._err is not None:
            raise self._err
        self._upload(slot)
        prev = self.slots[(self.k - 1) % self.depth]
        if prev["used"]:
            self.L.event_record(prev["free"], stream())   # everything queued so far on the training stream may still read `prev`
        self.L.stream_wait_event(stream(), slot["ready"])
        slot["used"] = True
        self.k = (self.k + 1) % self.depth
        self._start(self.slots[self.k])                     # decode the following epoch while this one trains
        pool = slot["pool"]
        return pool if slot["n"] == self.count else pool.rows(1, slot["n"])

    def close(self):
        global _prefetch_state
        _prefetch_state = None             # the pending pick dies with the loader
        if self._thread is not None:
            self._thread.join()
        self.L.stream_sync(self.copy_stream)
        for s_ in self.slots:
            self.L.host_free(s_["host"])
            if s_["dev_u8"]:
                self.L.free(s_["dev_u8"])
            if s_["side"]:
                self.L.host_free(s_["side"])
            self.L.event_destroy(s_["ready"]); self.L.event_destroy(s_["free"])
        self.L.stream_destroy(self.copy_stream)
        self.slots = []
