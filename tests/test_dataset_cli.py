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
