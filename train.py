#!/usr/bin/env python
"""train.lua on the engine: same flags (train.lua:15-49), same set-up order (:115-220: D, G,
criterion, flat parameters, optimiser state) and the same endless epoch loop (:223-248) around adversarial.train.
Per epoch (unless --noplot): the image grids of NN_UTILS.visualizeProgress (PNG files; no display server).  The V network
is not trained here (train_v.lua is out of scope): ratings appear only when a V is attached to the state.

    python train.py --batchSize 128 --N_epoch 1000 --epochs 3 --synthetic          # no dataset needed
    python train.py --dataDir dataset/out_aug_64x64 --colorSpace y --saveFreq 30
"""
import argparse
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    a = ap.add_argument
    a("--save", default="logs"); a("--saveFreq", type=int, default=30); a("--network", default="")
    a("--batchSize", type=int, default=32); a("--N_epoch", type=int, default=1000)
    a("--G_L1", type=float, default=0.0); a("--G_L2", type=float, default=0.0)
    a("--D_L1", type=float, default=0.0); a("--D_L2", type=float, default=1e-4)
    a("--D_iterations", type=int, default=1); a("--G_iterations", type=int, default=1)
    a("--D_maxAcc", type=float, default=1.01); a("--D_clamp", type=float, default=1.0); a("--G_clamp", type=float, default=5.0)
    a("--D_optmethod", default="adam"); a("--G_optmethod", default="adam")
    a("--D_sgd_lr", type=float, default=0.02); a("--G_sgd_lr", type=float, default=0.02)
    a("--D_sgd_momentum", type=float, default=0.0); a("--G_sgd_momentum", type=float, default=0.0)
    a("--gpu", type=int, default=0); a("--noiseDim", type=int, default=100); a("--scale", type=int, default=32)
    a("--seed", type=int, default=1); a("--colorSpace", default="rgb", choices=["rgb", "y"])
    a("--dataDir", default="dataset/out_aug_64x64"); a("--synthetic", action="store_true")
    a("--epochs", type=int, default=0, help="stop after this many epochs (0 = run forever, as train.lua does)")
    a("--noplot", action="store_true", help="train.lua:33 - skip the per-epoch image grids (logs/images*/<start>_<epoch>.png)")
    a("--blockingLoader", action="store_true", help="decode + upload each epoch's images on the training thread (dataset.loadRandomImages)")
    return ap.parse_args()


def main():
    o = parse()
    import torch
    cg = importlib.import_module("cat-generator_amd")
    torch.cuda.set_device(o.gpu)                                   # cutorch.setDevice(OPT.gpu + 1), train.lua:109
    cg.manual_seed(o.seed)                                         # train.lua:61-62,110
    C = 1 if o.colorSpace == "y" else 3
    IMG_DIMENSIONS = (C, o.scale, o.scale)                         # train.lua:74-78
    MODEL_D = cg.models.create_D(IMG_DIMENSIONS)                   # train.lua:147
    MODEL_G = cg.models.create_G(IMG_DIMENSIONS, o.noiseDim)       # train.lua:161
    print(MODEL_G); print(MODEL_D)
    print("Number of free parameters in D: %d" % cg.nn_utils.getNumberOfParameters(MODEL_D))
    print("Number of free parameters in G: %d" % cg.nn_utils.getNumberOfParameters(MODEL_G))
    S = cg.adversarial.State(vars(o), MODEL_G, MODEL_D)            # criterion, getParameters, OPTSTATE: :181-207
    ds = importlib.import_module("cat-generator_amd.dataset")
    ds.colorSpace = o.colorSpace; ds.setFileExtension("jpg"); ds.setHeight(o.scale); ds.setWidth(o.scale)
    ds.setDirs([o.dataDir]); ds.seed(o.seed)
    if o.network:   # after every generator was seeded: the checkpoint puts each of them back where the run stopped
        print(f"<trainer> reloading previously trained network: {o.network}")
        (cg.checkpoint.load_t7 if o.network.endswith(".net") else cg.checkpoint.load)(o.network, S)
    n_pool = o.N_epoch if o.N_epoch > 0 else 10000
    import time
    START_TIME = int(time.time())                                  # train.lua:58
    PLOT_DATA = []
    # train.lua:216: the same 100 noise vectors every epoch (a generator of their own: the training streams do not move)
    VIS_NOISE_INPUTS = np.random.RandomState(o.seed).uniform(-1, 1, (100, o.noiseDim)).astype(np.float32)
    loader = None if (o.synthetic or o.blockingLoader) else ds.AsyncLoader(n_pool)   # the next epoch's pool loads while this one trains
    while True:                                                    # train.lua:223
        print("Loading new training data...")
        if o.synthetic:
            pool = np.random.RandomState(S.EPOCH).rand(n_pool, C, o.scale, o.scale).astype(np.float32)
        elif loader is not None:
            pool = loader.next()
        else:
            pool = ds.loadRandomImages(n_pool).scaled              # train.lua:225
        TRAIN_DATA = cg.adversarial.TrainData(pool)
        if not o.noplot:                                           # train.lua:228-236
            first = cg.nn.as_nhwc(TRAIN_DATA.pool.rows(1, min(50, TRAIN_DATA.size()))).numpy()
            cg.nn_utils.visualizeProgress(S, VIS_NOISE_INPUTS, first, o.save, START_TIME, PLOT_DATA, verbose=True)
        cg.adversarial.train(S, TRAIN_DATA, o.D_maxAcc, max(20, min(1000 // o.batchSize, 250)))   # train.lua:238
        if (S.EPOCH - 1) % o.saveFreq == 0:                        # train.lua:241-244 (EPOCH was already advanced)
            os.makedirs(o.save, exist_ok=True)
            fn = os.path.join(o.save, "adversarial.npz")
            if os.path.exists(fn):
                os.replace(fn, fn + ".old")
            print(f"<trainer> saving network to {fn}")
            cg.checkpoint.save(fn, S)
            cg.checkpoint.export_t7(os.path.join(o.save, "adversarial.net"), S, PLOT_DATA)   # torch.save's format, train.lua:252-261
        if o.epochs and S.EPOCH > o.epochs:
            break


if __name__ == "__main__":
    main()
