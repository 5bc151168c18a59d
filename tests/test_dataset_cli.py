"""Host-side neighbours of the hot path: the dataset loader (dataset.lua) on CPU, the train.py
CLI on the GPU."""
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _make_jpgs(d, n=6, size=64):
    from PIL import Image
    rs = np.random.RandomState(0)
    for i in range(n):
        Image.fromarray((rs.rand(size, size, 3) * 255).astype(np.uint8)).save(os.path.join(d, f"cat_{i:03d}.jpg"), quality=95)


def test_dataset_loader(tmp_path):
    ds = importlib.import_module("cat-generator_amd.dataset")
    _make_jpgs(str(tmp_path))
    (tmp_path / "notes.txt").write_text("ignored")
    ds.setDirs([str(tmp_path)]); ds.setFileExtension("jpg"); ds.setHeight(32); ds.setWidth(32); ds.seed(1)
    ds.colorSpace = "rgb"
    assert len(ds.loadPaths()) == 6 and ds.paths == sorted(ds.paths)
    d = ds.loadRandomImages(4)
    assert d.size() == 4 and d.scaled.shape == (4, 3, 32, 32) and d.scaled.dtype == np.float32
    assert d.scaled.min() >= 0.0 and d.scaled.max() <= 1.0
    ds.colorSpace = "y"
    y = ds.loadRandomImages(100)  # more than available -> all of them (dataset.lua:162)
    assert y.scaled.shape == (6, 1, 32, 32)
    rgb = np.random.RandomState(1).rand(3, 4, 4).astype(np.float32)
    np.testing.assert_allclose(ds.rgb2y(rgb)[0], 0.21 * rgb[0] + 0.72 * rgb[1] + 0.07 * rgb[2], rtol=1e-6)
    ds.colorSpace = "rgb"


@pytest.mark.parametrize("Hs,Ws,h,w", [(64, 64, 32, 32), (64, 64, 64, 64), (32, 32, 64, 64), (96, 60, 32, 24), (80, 80, 32, 32), (20, 16, 32, 32)])
def test_image_scale_follows_the_oracle_rule(Hs, Ws, h, w):
    """dataset.image_scale (numpy, what loadRandomImages applies to image.load's floats, dataset.lua:129-131) against the oracle's
    scalar restatement of the `image` rock's separable scaling: bit-equal for shrinking by 2 / 3 / 2.5 / 1.25, copying and enlarging;
    plus the hand-computable cases."""
    from oracle import oracle as O
    ds = importlib.import_module("cat-generator_amd.dataset")
    rs = np.random.RandomState(Hs + w)
    img = (rs.randint(0, 256, size=(3, Hs, Ws)).astype(np.float32) / np.float32(255))
    np.testing.assert_array_equal(ds.image_scale(img, w, h), O.image_scale(img, w, h))
    const = np.full((3, Hs, Ws), np.float32(0.3), np.float32)
    np.testing.assert_allclose(ds.image_scale(const, w, h), 0.3, rtol=2e-7)       # averages of a constant
    if (Hs, Ws, h, w) == (64, 64, 32, 32):   # exact 2x: the 2x2 box mean, rows first: ((a + b)/2 + (c + d)/2)/2
        f = np.float32
        a, b, c, d = img[:, 0::2, 0::2], img[:, 0::2, 1::2], img[:, 1::2, 0::2], img[:, 1::2, 1::2]
        np.testing.assert_array_equal(ds.image_scale(img, 32, 32), ((a + b) / f(2) + (c + d) / f(2)) / f(2))
    if h > Hs:     # enlarging keeps the corner samples
        out = ds.image_scale(img, w, h)
        assert np.array_equal(out[:, 0, 0], img[:, 0, 0]) and np.array_equal(out[:, -1, -1], img[:, -1, -1])


def test_checkpoint_during_a_pending_prefetch_resumes_on_the_same_pools(tmp_path):
    """ADVICE round 2 (train.py:68): AsyncLoader has already drawn epoch E+1's permutation when epoch E ends and a resumed loader
    draws again on construction.  checkpoint.save stores dataset.checkpoint_state() - the generator's state from BEFORE the pending
    pick - so the resumed run trains E+1, E+2 ... on exactly the pools the uninterrupted (or blocking-loader) run uses."""
    ds = importlib.import_module("cat-generator_amd.dataset")
    ck = importlib.import_module("cat-generator_amd.checkpoint")
    _make_jpgs(str(tmp_path), n=9)
    ds.setDirs([str(tmp_path)]); ds.setFileExtension("jpg"); ds.seed(3)
    blocking = [ds._pick(5) for _ in range(4)]                     # epochs 1..4 with --blockingLoader
    ds.seed(3)
    pending = ds._prefetch_pick(5)                                 # AsyncLoader.__init__: epoch 1's pool starts loading
    assert pending == blocking[0]
    consumed, pending = pending, ds._prefetch_pick(5)              # next(): epoch 1 trains, epoch 2 prefetches
    out = {}
    ck._pack_rs("dataset_random", ds.checkpoint_state(), out)      # checkpoint.save after epoch 1
    np.savez(tmp_path / "c.npz", **out)
    ds.seed(77)                                                    # a new process: train.py seeds, then loads
    ck._unpack_rs("dataset_random", np.load(tmp_path / "c.npz"), ds.restore_state)
    assert ds._prefetch_pick(5) == blocking[1]                     # the resumed loader's first pool is epoch 2's
    assert ds._prefetch_pick(5) == blocking[2]
    ds.seed(3)                                                     # blocking loader: nothing pending, the current state is stored
    ds._pick(5)
    st = ds.checkpoint_state()
    ds.seed(78); ds.restore_state(st)
    assert ds._pick(5) == blocking[1]
    ds.seed(1)


def test_image_grids_and_png_writer(tmp_path):
    """nn_utils.lua:526-583: grid layout (row by row, 7 extra rows), the epoch digits at the bottom right (3 x 5 glyphs, last
    digit rightmost, 6 px pitch), and the PNG on disk read back with PIL."""
    from PIL import Image
    U = importlib.import_module("cat-generator_amd.nn_utils")
    rs = np.random.RandomState(0)
    imgs = rs.rand(7, 3, 8, 8).astype(np.float32)
    g = U.imagesToGridTensor(imgs, 2, 3, 407)
    assert g.shape == (3, 2 * 8 + 7, 3 * 8)
    np.testing.assert_array_equal(g[:, 0:8, 8:16], imgs[1])
    np.testing.assert_array_equal(g[:, 8:16, 0:8], imgs[3])       # second row starts with the 4th image; the 7th is dropped
    Hpx, Wpx = g.shape[1:]
    seven = np.array([[1, 1, 1], [0, 0, 1], [0, 0, 1], [0, 0, 1], [0, 0, 1]], np.float32)
    four = np.array([[1, 0, 1], [1, 0, 1], [1, 1, 1], [0, 0, 1], [0, 0, 1]], np.float32)
    zero = np.array([[1, 1, 1], [1, 0, 1], [1, 0, 1], [1, 0, 1], [1, 1, 1]], np.float32)
    for pos, glyph in ((1, seven), (2, zero), (3, four)):
        x0 = Wpx - 1 - pos * 5 - pos - 1
        np.testing.assert_array_equal(g[1, Hpx - 7:Hpx - 2, x0:x0 + 3], glyph)
    assert g[:, Hpx - 2:, :].sum() == 0 and Hpx - 7 == 16              # digits start right below the images, two blank rows under them
    path = tmp_path / "images" / "0_00407.png"
    U.saveImagesAsGrid(str(path), imgs, 2, 3, 407)
    back = np.asarray(Image.open(str(path)))
    assert back.shape == (Hpx, Wpx, 3)
    np.testing.assert_array_equal(back.transpose(2, 0, 1), np.rint(g * 255).astype(np.uint8))
    y = U.toRgb(rs.rand(2, 1, 8, 8).astype(np.float32), "y")
    assert y.shape == (2, 3, 8, 8) and np.array_equal(y[:, 0], y[:, 2])
    gray = U._png_bytes(rs.rand(1, 5, 4).astype(np.float32))        # single-channel PNGs too
    import io
    assert np.asarray(Image.open(io.BytesIO(gray))).shape == (5, 4)


@pytest.mark.gpu
@pytest.mark.parametrize("cs", ["rgb", "y"])
def test_async_loader_pools_equal_the_blocking_loader(tmp_path, cs):
    """dataset.AsyncLoader (pinned 8-bit staging, copy stream, device-side conversion, two HBM pools) against
    loadRandomImages + upload: the same images in the same order with the same bits, over four epochs (both pools reused),
    with training-stream work reading the previous pool in between, and when fewer files exist than were asked for."""
    import torch
    cg = importlib.import_module("cat-generator_amd")
    ds = importlib.import_module("cat-generator_amd.dataset")
    from oracle import oracle as O
    _make_jpgs(str(tmp_path), n=9)
    ds.setDirs([str(tmp_path)]); ds.setFileExtension("jpg"); ds.setHeight(32); ds.setWidth(32)
    ds.colorSpace = cs
    try:
        # the device kernel (image.load -> image.scale -> colour space from the decoded bytes) against the oracle, several geometries
        import ctypes
        rs = np.random.RandomState(3)
        for (Hs, Ws, h, w) in ((64, 64, 32, 32), (64, 64, 64, 64), (32, 32, 64, 64), (96, 60, 32, 24), (80, 80, 32, 32)):
            u8 = rs.randint(0, 256, size=(2, Hs, Ws, 3)).astype(np.uint8)
            src = torch.from_numpy(u8).cuda()
            dst = cg.Tensor.empty((2, 1 if cs == "y" else 3, h, w), "nhwc")
            cg.lib().images_u8_scale_to_f32(cg.tensor.stream(), src.data_ptr(), dst.ptr, 2, Hs, Ws, h, w, 1 if cs == "y" else 0)
            want = np.stack([O.load_image(u8[i], w, h, cs) for i in range(2)])
            np.testing.assert_array_equal(dst.numpy(), want, err_msg=f"{Hs}x{Ws} -> {h}x{w} {cs}")
        ds.seed(5)
        ref = [ds.loadRandomImages(6).scaled for _ in range(4)]
        ds.seed(5)
        ld = ds.AsyncLoader(6)
        keep = []
        for e in range(4):
            pool = ld.next()
            data = cg.adversarial.TrainData(pool)
            assert data.size() == 6
            np.testing.assert_array_equal(cg.nn.as_nhwc(pool).numpy(), ref[e])
            keep.append(pool.t.sum())            # work on the training stream that reads this pool while the next one loads
        torch.cuda.synchronize()
        ld.close()
        ds.seed(7)
        few_ref = ds.loadRandomImages(20).scaled
        ds.seed(7)
        ld = ds.AsyncLoader(20)
        few = ld.next()
        assert few.shape[0] == 9
        np.testing.assert_array_equal(cg.nn.as_nhwc(few).numpy(), few_ref)
        ld.close()
        # a dataset of MIXED sizes (dataset.lua:129-131 scales every image on its own): three files of another size among the nine
        # take the host path inside the loader and land in the pool with the blocking loader's bits
        from PIL import Image
        rs = np.random.RandomState(11)
        for k, (hh, ww) in enumerate(((48, 80), (64, 40), (100, 100))):
            Image.fromarray(rs.randint(0, 256, size=(hh, ww, 3)).astype(np.uint8)).save(os.path.join(str(tmp_path), f"odd{k}.jpg"), quality=95)
        ds.setDirs([str(tmp_path)])
        ds.seed(9)
        mixed_ref = [ds.loadRandomImages(12).scaled for _ in range(2)]
        ds.seed(9)
        ld = ds.AsyncLoader(12)
        for e in range(2):
            np.testing.assert_array_equal(cg.nn.as_nhwc(ld.next()).numpy(), mixed_ref[e])
        ld.close()
        # sources more than 6x the target (the device kernel's box loop stops there): every image takes the host path; since round 5 the
        # loader then pins an fp32 buffer of the POOL's size instead of an 8-bit one of the source size, and uploads it in one copy
        big = tmp_path / "big"
        os.makedirs(str(big))
        for k in range(5):
            Image.fromarray(rs.randint(0, 256, size=(200, 208, 3)).astype(np.uint8)).save(os.path.join(str(big), f"b{k}.jpg"), quality=95)
        ds.setDirs([str(big)])
        ds.seed(13)
        big_ref = [ds.loadRandomImages(4).scaled for _ in range(3)]
        ds.seed(13)
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ld = ds.AsyncLoader(4)
        assert ld.host_all and ld.nbytes == 4 * 32 * 32 * (1 if cs == "y" else 3) * 4
        for e in range(3):
            np.testing.assert_array_equal(cg.nn.as_nhwc(ld.next()).numpy(), big_ref[e])
        ld.close()
    finally:
        ds.colorSpace = "rgb"


@pytest.mark.gpu
def test_train_cli_runs_epochs_and_resumes(tmp_path):
    _make_jpgs(str(tmp_path), n=40)
    cmd = [sys.executable, os.path.join(ROOT, "train.py"), "--batchSize", "16", "--N_epoch", "32", "--epochs", "2",
           "--dataDir", str(tmp_path), "--save", str(tmp_path / "logs"), "--saveFreq", "1", "--colorSpace", "y"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "Number of free parameters in D: 6663337" in out.stdout     # C=1
    assert out.stdout.count("<trainer> Epoch #") == 2 and "Confusion of D:" in out.stdout
    ck = tmp_path / "logs" / "adversarial.npz"
    assert ck.exists() and (tmp_path / "logs" / "adversarial.npz.old").exists()
    out2 = subprocess.run(cmd + ["--network", str(ck), "--epochs", "3"], capture_output=True, text=True, timeout=600)
    assert out2.returncode == 0, out2.stderr[-2000:]
    assert "<trainer> Epoch #3" in out2.stdout and "<trainer> Epoch #1 " not in out2.stdout


def test_display_grid_and_neighbour_helpers():
    """sample.lua's helpers: image.toDisplayTensor's layout (nrow images per row, global min-max), randperm selection,
    nearest training-set neighbour."""
    U = importlib.import_module("cat-generator_amd.nn_utils")
    rs = np.random.RandomState(3)
    imgs = (rs.rand(5, 3, 4, 4) * 0.5 + 0.25).astype(np.float32)
    g = U.toDisplayTensor(imgs, 2)
    assert g.shape == (3, 3 * 4, 2 * 4) and g.min() == 0.0 and g.max() == 1.0
    lo, hi = 0.0, imgs.max()                      # the unused cell is black, so the minimum of the grid is 0
    np.testing.assert_allclose(g[:, 4:8, 0:4], (imgs[2] - lo) / (hi - lo), rtol=1e-6)
    sel = U.selectRandomImagesFrom(imgs, 3, np.random.RandomState(1))
    perm = np.random.RandomState(1).permutation(5)
    np.testing.assert_array_equal(sel, imgs[perm[:3]])
    train = rs.rand(7, 3, 4, 4).astype(np.float32)
    pairs = U.findClosestNeighboursOf(train[[4, 1]] + 1e-3, train)
    assert np.array_equal(pairs[0][1], train[4]) and np.array_equal(pairs[1][1], train[1]) and pairs[0][2] < 0.01


@pytest.mark.gpu
def test_sample_cli_writes_the_grids_from_a_torch7_checkpoint(tmp_path):
    """train.py (one epoch, saves adversarial.net in torch.save's format) -> sample.py: the seven grids of sample.lua:77-121 with
    their sizes (1024 samples in 32 x 32 cells, 64 in 8 x 8, 16 neighbour pairs in two rows)."""
    from PIL import Image
    _make_jpgs(str(tmp_path), n=40)
    logs = tmp_path / "logs"
    cmd = [sys.executable, os.path.join(ROOT, "train.py"), "--batchSize", "16", "--N_epoch", "32", "--epochs", "1", "--noplot",
           "--dataDir", str(tmp_path), "--save", str(logs), "--saveFreq", "1"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert (logs / "adversarial.net").exists()
    dst = tmp_path / "samples"
    cmd = [sys.executable, os.path.join(ROOT, "sample.py"), "--save", str(logs), "--writeto", str(dst), "--dataDir", str(tmp_path),
           "--neighbours", "--batchSize", "64"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    sizes = {"trainset_s1_0001_base.jpg": (5 * 32, 8 * 32), "random256_0001_base.jpg": (16 * 32, 16 * 32),
             "random1024_0001_base.jpg": (32 * 32, 32 * 32), "best_0001_base.jpg": (8 * 32, 8 * 32),
             "worst_0001_base.jpg": (8 * 32, 8 * 32), "random_0001_base.jpg": (8 * 32, 8 * 32),
             "best_0001_neighbours_base.jpg": (2 * 32, 16 * 32)}
    for name, (h, w) in sizes.items():
        im = np.asarray(Image.open(str(dst / name)))
        assert im.shape == (h, w, 3), (name, im.shape)
