"""Data-parallel plumbing on the GPU box: two ranks (sharing the single MI355X gpurun provides) run the product's
DP iteration — sync-BN all-reduces inside G, all-reduce(mean) of the flat gradients before the fused
penalty/clamp/Adam — and must stay bit-identical replicas with finite parameters.  The DP *maths* (sharded ==
full batch) is pinned on CPU in tests/test_dp_gloo.py; 8-GPU RCCL runs belong to the driver."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, backend, overlap, q):
    try:
        sys.path.insert(0, ROOT)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          LOCAL_RANK="0")
        torch.cuda.set_device(0)
        dist.init_process_group(backend, rank=rank, world_size=world)
        cg = importlib.import_module("cat-generator_amd")
        cg.parallel.attach(world, rank)
        cg.manual_seed(7)
        G, D = cg.models.create_G((3, 32, 32), 100), cg.models.create_D((3, 32, 32))
        S = cg.adversarial.State(dict(batchSize=8, seed=1 + rank, overlap_comm=overlap), G, D)
        cg.tensor.rng().offset += rank << 40
        pool = np.random.RandomState(100 + rank).rand(32, 3, 32, 32).astype(np.float32)
        data = cg.adversarial.TrainData(pool)
        for _ in range(2):
            cg.adversarial.iteration(S, data)
        torch.cuda.synchronize()
        q.put((rank, S.PARAMETERS_G.numpy(), S.PARAMETERS_D.numpy(), None))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # report instead of hanging the parent
        q.put((rank, None, None, f"{type(e).__name__}: {e}"))


def _run(backend, overlap):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + os.getpid() % 200 + (50 if overlap else 0)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, backend, overlap, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=400) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(60)
    errs = [r[3] for r in res if r[3]]
    if errs and ("gloo" in errs[0].lower() or "not supported" in errs[0].lower() or "cuda" in errs[0].lower()):
        pytest.skip(f"{backend} cannot all-reduce device tensors on this box: {errs[0][:200]}")
    assert not errs, errs
    (_, g0, d0, _), (_, g1, d1, _) = res
    assert np.isfinite(g0).all() and np.isfinite(d0).all()
    np.testing.assert_array_equal(g0, g1)
    np.testing.assert_array_equal(d0, d1)
    return g0, d0


@pytest.mark.timeout(900)
def test_two_ranks_stay_replicas_and_overlap_is_result_neutral():
    """Two ranks on the GPU box through the product's DP iteration: replicas stay bit-equal, with the blocking
    collectives and with the overlapped ones (D's all-reduce under the G-step's generator forward, G's all-reduce in
    buckets under G's backward); both schedules give the same parameters up to the usual atomics-order drift."""
    g_o, d_o = _run("gloo", True)
    g_b, d_b = _run("gloo", False)
    for a, b in ((g_o, g_b), (d_o, d_b)):
        d = np.abs(a - b)
        assert d.max() <= 2 * 2.5e-3 and d.mean() <= 2e-5, (d.max(), d.mean())
