"""catgan engine for MI355X (gfx950): the G+D training step of aleju/cat-generator behind Torch7's nn API.

    import importlib; cg = importlib.import_module("cat-generator_amd")
    G = cg.models.create_G((3, 32, 32), 100); D = cg.models.create_D((3, 32, 32))
    S = cg.adversarial.State({"batchSize": 128}, G, D)
    cg.adversarial.iteration(S, cg.adversarial.TrainData(real_pool))

The arithmetic lives in lib/libcatgan_hip.so (csrc/*.hip, C ABI in include/catgan.h).  There is no CPU or
PyTorch fallback for it: a missing library or a failing HIP call raises.
"""
from . import _abi, adversarial, checkpoint, cudnn, models, nn, nn_utils, optim, parallel, tensor, weight_init  # noqa: F401
from ._abi import CatganError, lib  # noqa: F401
from .tensor import Tensor, manual_seed  # noqa: F401

__all__ = ["nn", "cudnn", "optim", "models", "adversarial", "nn_utils", "parallel", "Tensor", "manual_seed", "lib"]
