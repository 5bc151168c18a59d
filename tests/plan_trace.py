"""Helpers for the planner tests: run a network through cg_net_* in TRACE mode (no GPU: launches go to recording stubs,
csrc/net_ktable.inc) and canonicalise the text so that it does not depend on addresses."""
import importlib
import re

import numpy as np
import torch


def cg_pkg():
    return importlib.import_module("cat-generator_amd")


def build(which, seed=3):
    cg = cg_pkg()
    cg.manual_seed(seed)
    if which == "G32up-c":
        return cg.models.create_G((3, 32, 32), 100), (100,), "plain"
    if which == "G32up":
        return cg.models.create_G_decoder_upsampling32((1, 32, 32), 100), (100,), "plain"
    if which == "G32up-c@64":
        return cg.models.create_G((3, 64, 64), 100), (100,), "plain"
    if which == "D32_st3":
        return cg.models.create_D((3, 32, 32)), (3, 32, 32), "nhwc"
    if which == "D32_st3@64":
        return cg.models.create_D((3, 64, 64)), (3, 64, 64), "nhwc"
    raise KeyError(which)


def trace(which, N, options=(), dp=None, rng_offset=1000):
    """-> dict(forward=[lines], backward=[...], updateGradInput=[...], draws=int, out_shape=..., net=PlannedNet)."""
    cg = cg_pkg()
    P = importlib.import_module("cat-generator_amd.planned")
    net, ishape, fmt = build(which)
    net.getParameters()
    x = cg.Tensor(torch.zeros(N * int(np.prod(ishape))), (N,) + ishape, fmt)
    pn = P.PlannedNet(net, trace=True)
    L = cg.lib()
    for k, v in options:
        L.net_set_option(pn.h, k.encode(), int(v))
    if dp:
        net._bucket_overlap = dp.get("buckets", False)
        cg.parallel.attach(dp["world"], 0)
    try:
        L.net_trace_region(pn.h, x.ptr, x.t.numel() * 4)
        r = cg.tensor.rng()
        r.offset = rng_offset
        pn.forward(x)
        first = pn.take_trace().strip().split("\n")   # first pass: weight packing included
        r.offset = rng_offset
        y = pn.forward(x)
        res = dict(first_forward=first, forward=pn.take_trace().strip().split("\n"), draws=r.offset - rng_offset, out_shape=y.shape,
                   out_fmt=y.fmt)
        gy = cg.Tensor(torch.zeros(int(np.prod(y.shape))), y.shape, y.fmt)
        L.net_trace_region(pn.h, gy.ptr, max(gy.t.numel() * 4, 4))
        gi = pn.backward(x, gy, True)
        res["backward"] = pn.take_trace().strip().split("\n")
        pn.backward(x, gy, False)
        res["updateGradInput"] = pn.take_trace().strip().split("\n")
        res["gin_shape"] = gi.shape
        res["stats"] = pn.stats()
        res["net"], res["module"] = pn, net
        return res
    finally:
        if dp:
            cg.parallel.attach(1, 0)


def protos():
    return cg_pkg().lib().protos


def canon(lines):
    """Workspace arguments dropped, floats normalised, regions renumbered by first appearance: two traces are equal iff the same
    entry points run in the same order on the same geometry with the same data flow."""
    names = {}
    P = protos()

    def rn(m):
        k = m.group(1)
        if k not in names:
            names[k] = f"R{len(names)}"
        return names[k] + "+" + m.group(2)

    out = []
    for l in lines:
        f = l.split("|")
        if f[0] != "call":
            out.append(l)
            continue
        keep = []
        for (t, an), v in zip(P[f[1]][1], f[2:]):
            if an in ("ws", "ws_bytes"):
                continue
            keep.append(("f:" + repr(float.fromhex(v[2:]))) if v.startswith("f:") else v)
        out.append(re.sub(r"r(\d+)\+(\d+)", rn, "|".join(["call", f[1]] + keep)))
    return out


def calls(lines, name=None):
    """[(entry point, {argname: token})] of the launch lines (optionally one entry point only)."""
    P = protos()
    out = []
    for l in lines:
        f = l.split("|")
        if f[0] != "call" or (name and f[1] != name):
            continue
        out.append((f[1], {an: v for (t, an), v in zip(P[f[1]][1], f[2:])}))
    return out
