"""world_size-2 data-parallel test on CPU (gloo): the exchange steps of cat-generator_amd/parallel.py —
all-reduce(mean) of the flat gradient BEFORE penalty/clamp/Adam, and sync-BN statistics — make two ranks on
half batches reproduce the single-process step on the full batch.  Compute is the CPU oracle (test
infrastructure); the collectives and their ordering are the product's."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _make(seed):
    from oracle import oracle as O
    rng = O.RNG(seed)
    G = O.create_G32up_c(3, 100, rng)
    D = O.create_D32_st3(3, 32, rng)
    return O, G, D


def _fix_masks(O, D, N, seed):
    rs = np.random.RandomState(seed)
    masks = []
    for m in D.modules():
        if isinstance(m, O.SpatialDropout):
            masks.append((m, (rs.rand(N, 400) < (1 - m.p)).astype(np.float32)))
        elif isinstance(m, O.Dropout):
            masks.append((m, (rs.rand(N, 256) < 0.5).astype(np.float32) * 2))
    return masks


def _set_masks(masks, rows, widths):
    for (m, full), w in zip(masks, widths):
        m.fixed = np.ascontiguousarray(full[rows, :w])


def _run_step(O, G, D, x, t, z, hooks=None):
    """fevalD-like pass on D then fevalG-like pass through G (adversarial.lua:72-112,171-215) on the given shard."""
    T = O.Trainer(G, D)
    if hooks:
        T.grad_allreduce = hooks["grad"]
        for m in G.modules():
            if isinstance(m, O.SBN):
                m.stat_allreduce = hooks["bn"]
    T.feval_D(x, t)
    O.adam(T.pD, T.gD, T.stD)
    T.feval_G(z, np.ones(z.shape[0], np.float32))
    O.adam(T.pG, T.gG, T.stG)
    return T


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    par = importlib.import_module("cat-generator_amd.parallel")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    par.attach(world, rank)
    O, G, D = _make(5)
    O.set_num_threads(2)
    N = 4
    rs = np.random.RandomState(0)
    x = rs.rand(N, 3, 32, 32).astype(np.float32); t = (rs.rand(N) > 0.5).astype(np.float32)
    z = (rs.rand(N, 100) * 2 - 1).astype(np.float32)
    rows = slice(rank * N // world, (rank + 1) * N // world)
    masks = _fix_masks(O, D, N, 3)
    chans = [64, 64, 64, 64, 128, 320, 256]
    _set_masks(masks, rows, chans)

    def grad_hook(g):  # product collective: mean over ranks, in place on the flat vector
        tt = torch.from_numpy(g)
        par.allreduce_mean_(tt)

    def bn_hook(s1, s2, cnt):  # product collective: sum of fp64 statistics; count scales with world size
        buf = torch.from_numpy(np.concatenate([s1, s2]))
        par.allreduce_sum_(buf)
        k = s1.size
        return buf[:k].numpy().copy(), buf[k:].numpy().copy(), cnt * par.world_size()

    assert par.sync_bn_active()
    # host-scalar reduction behind the accuracy gate / confusion counts: identical global values on every rank
    hits, tot = par.allreduce_sum_host([float(rank + 1), 2.0])
    assert (hits, tot) == (3.0, 4.0)
    T = _run_step(O, G, D, x[rows], t[rows], z[rows], dict(grad=grad_hook, bn=bn_hook))
    q.put((rank, T.pD.copy(), T.pG.copy(), T.stD["m"].copy(), T.stG["m"].copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_equal_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + os.getpid() % 200
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # single-process reference on the full batch with the same masks
    O, G, D = _make(5)
    N = 4
    rs = np.random.RandomState(0)
    x = rs.rand(N, 3, 32, 32).astype(np.float32); t = (rs.rand(N) > 0.5).astype(np.float32)
    z = (rs.rand(N, 100) * 2 - 1).astype(np.float32)
    masks = _fix_masks(O, D, N, 3)
    _set_masks(masks, slice(0, N), [64, 64, 64, 64, 128, 320, 256])
    T = _run_step(O, G, D, x, t, z)
    (_, pD0, pG0, mD0, mG0), (_, pD1, pG1, mD1, mG1) = res
    np.testing.assert_array_equal(pD0, pD1)  # ranks stay replicas
    np.testing.assert_array_equal(pG0, pG1)
    # Adam's first moment = (1-b1) * clamped mean gradient: the sharded gradient equals the full-batch one
    for name, a, b in (("mD", mD0, T.stD["m"]), ("mG", mG0, T.stG["m"])):
        scale = np.abs(b).max()
        d = np.abs(a - b)
        assert d.max() <= 3e-2 * scale and d.mean() <= 1e-3 * scale, (name, d.max(), d.mean(), scale)
    for name, a, b in (("pD", pD0, T.pD), ("pG", pG0, T.pG)):
        d = np.abs(a - b)
        assert d.max() <= 2.5e-3 and d.mean() <= 2e-5, (name, d.max(), d.mean())
