#!/usr/bin/env python
"""Generates tests/golden/step_kat.npz — known-answer vectors of the hot path, produced by the CPU oracle
(oracle/oracle.py; the reference itself cannot run here and ships no vectors).  Regenerate only
deliberately:  python tests/golden/make_golden.py

Contents (seed 77, batch 4, G32up-c / D32_st3, 32x32 RGB, explicit real batch / noise, masks from the shared
counter stream):
  real, noise_d, noise_g      the injected inputs of two consecutive iterations
  fake_k, outD_k, outG_k      generator images of the D-step, D outputs of the D-step / G-step, iteration k
  pG_probe_k, pD_probe_k      parameters at 4096 fixed probe indices after iteration k
  pG_sum_k, pD_sum_k          float64 sums of the flat parameter vectors after iteration k
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

SEED, N, STEPS = 77, 4, 2


def run():
    rng = O.RNG(SEED)
    T = O.Trainer(O.create_G32up_c(3, 100, rng), O.create_D32_st3(3, 32, rng))
    rs = np.random.RandomState(SEED)
    out = {}
    probe_g = rs.randint(0, T.pG.size, size=4096)
    probe_d = rs.randint(0, T.pD.size, size=4096)
    out["probe_g"], out["probe_d"] = probe_g, probe_d
    reals, nds, ngs = [], [], []
    for k in range(STEPS):
        real = rs.rand(N // 2, 3, 32, 32).astype(np.float32)
        nd = (rs.rand(N // 2, 100) * 2 - 1).astype(np.float32)
        ng = (rs.rand(N, 100) * 2 - 1).astype(np.float32)
        r = T.step(real, nd, ng)
        reals.append(real); nds.append(nd); ngs.append(ng)
        out[f"fake_{k}"] = r["fake"]
        out[f"outD_{k}"] = r["outD"]
        out[f"outG_{k}"] = r["outG"]
        out[f"pG_probe_{k}"] = T.pG[probe_g].copy()
        out[f"pD_probe_{k}"] = T.pD[probe_d].copy()
        out[f"pG_sum_{k}"] = np.float64(T.pG.sum(dtype=np.float64))
        out[f"pD_sum_{k}"] = np.float64(T.pD.sum(dtype=np.float64))
    out["real"], out["noise_d"], out["noise_g"] = np.stack(reals), np.stack(nds), np.stack(ngs)
    return out


if __name__ == "__main__":
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "step_kat.npz")
    np.savez_compressed(path, **run())
    print("wrote", path, os.path.getsize(path), "bytes")
