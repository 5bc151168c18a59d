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


def _worker(rank, world, port, backend, overlap, q, max_acc=1.01):
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
        d_before = S.PARAMETERS_D.numpy().copy()
        for _ in range(2):
            cg.adversarial.iteration(S, data, maxAccuracyD=max_acc)
        torch.cuda.synchronize()
        if max_acc <= 0.0:   # the gate held D back on every rank, G trained on
            assert np.array_equal(S.PARAMETERS_D.numpy(), d_before), "D moved although the accuracy gate was closed"
        q.put((rank, S.PARAMETERS_G.numpy(), S.PARAMETERS_D.numpy(), None))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # report instead of hanging the parent
        q.put((rank, None, None, f"{type(e).__name__}: {e}"))


def _run(backend, overlap, max_acc=1.01):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + os.getpid() % 200 + (50 if overlap else 0) + (25 if max_acc <= 1 else 0)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, backend, overlap, q, max_acc)) for r in range(2)]
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


@pytest.mark.timeout(900)
def test_closed_accuracy_gate_under_overlapped_collectives():
    """adversarial.lua:150-166 under DP: with the gate closed (D_maxAcc = 0) fevalD returns false, false AFTER the
    overlapped all-reduce of D's gradient was started - every rank must take the same decision (global-batch accuracy),
    finish the collective, leave D untouched and keep training G; the replicas stay bit-equal."""
    g, d = _run("gloo", True, max_acc=0.0)
    assert np.isfinite(g).all() and np.isfinite(d).all()


# ------------------------------------------------------------------ 2 ranks x N/2 == 1 rank x N on the HIP path
def _shard_worker(rank, world, port, overlap, q, comm="torch"):
    """world == 1: the full batch in one process; world == 2: rank r takes rows [r*N/2R, (r+1)*N/2R) of the real / noise
    draws (SURVEY.md 8e partitioning).  Dropout probabilities are set to 0 so that no mask depends on the position of a
    sample in the counter stream - everything else (sync-BN, gradient buckets, overlap, fused penalty / clamp / Adam) is
    the product's DP iteration on the HIP kernels."""
    try:
        sys.path.insert(0, ROOT)
        torch.cuda.set_device(0)
        os.environ["CG_COMM"] = comm
        if world > 1:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
            dist.init_process_group("gloo", rank=rank, world_size=world)
        cg = importlib.import_module("cat-generator_amd")
        if world > 1:
            cg.parallel.attach(world, rank)
            want = "abi1+torch" if comm == "abi1" else "torch"
            assert cg.parallel.comm_backend() == want, (cg.parallel.comm_backend(), want)
            if comm == "abi1":   # both communicators exist, sized 1, and RCCL reports its version
                info = cg.parallel.comm_info()
                assert info["comm_nranks"] == 1 and info["rccl_version"] > 20000, info
        cg.manual_seed(11)
        G, D = cg.models.create_G((3, 32, 32), 100), cg.models.create_D((3, 32, 32))
        for net in (G, D):
            for m in net.listModules():
                if isinstance(m, (cg.nn.SpatialDropout, cg.nn.Dropout)):
                    m.p = 0.0
        N = 16
        S = cg.adversarial.State(dict(batchSize=N // world, seed=1, overlap_comm=overlap), G, D)
        S.keep_outputs = True
        rs = np.random.RandomState(3)
        pool = rs.rand(24, 3, 32, 32).astype(np.float32)
        data = cg.adversarial.TrainData(pool)
        out = []
        for it in range(2):
            idx = rs.randint(0, 24, size=N // 2)
            zD = (rs.rand(N // 2, 100) * 2 - 1).astype(np.float32)
            zG = (rs.rand(N, 100) * 2 - 1).astype(np.float32)
            h, n = N // 2 // world, N // world
            cg.adversarial.iteration(S, data, N // world, real_idx=idx[rank * h:(rank + 1) * h],
                                     noise_D=zD[rank * h:(rank + 1) * h], noise_G=zG[rank * n:(rank + 1) * n])
            out.append((S._last["gD"].numpy().copy(), S._last["gG"].numpy().copy()))
        torch.cuda.synchronize()
        q.put((rank, out, S.PARAMETERS_G.numpy(), S.PARAMETERS_D.numpy(), None))
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
    except Exception as e:
        import traceback
        q.put((rank, None, None, None, f"{type(e).__name__}: {e}\n{traceback.format_exc()[-1500:]}"))


def _run_sharded(world, overlap, comm="torch"):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29950 + os.getpid() % 40 + (5 if overlap else 0) + (11 if comm != "torch" else 0)
    procs = [ctx.Process(target=_shard_worker, args=(r, world, port, overlap, q, comm)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=400) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(60)
    errs = [r[4] for r in res if r[4]]
    assert not errs, errs
    return res


@pytest.mark.timeout(900)
@pytest.mark.parametrize("overlap,comm", [(False, "torch"), (True, "torch"), (True, "abi1")])
def test_two_ranks_on_half_batches_equal_one_rank_on_the_full_batch(overlap, comm):
    """with the global batch split over two ranks (sync-BN statistics and the mean of the flat gradients
    exchanged; `overlap`: D's all-reduce under the G-step's generator forward, G's in buckets under its backward) the
    gradients handed to Adam and the parameters after two iterations equal the single-rank run on the full batch, up
    to fp32 re-association of the batch reductions.  Both ranks run the HIP kernels on the one GPU of this box; the
    transport is gloo (RCCL refuses two ranks on one device) - the collectives' placement and order are the product's.
    comm = "abi1" (CG_COMM=abi1): every exchange ALSO goes through the cg_comm_* entry points on single-rank communicators
    (one for the gradients, one for the sync-BN sums: fork / join events, the buckets cg_net_backward starts, _PendingAbi), so that
    code runs here although its cross-rank arithmetic needs a multi-GPU node."""
    full = _run_sharded(1, overlap)[0]
    r0, r1 = _run_sharded(2, overlap, comm)
    np.testing.assert_array_equal(r0[2], r1[2]); np.testing.assert_array_equal(r0[3], r1[3])   # replicas
    for it in range(2):
        for k, name in ((0, "D"), (1, "G")):
            a, b = r0[1][it][k], full[1][it][k]
            scale = np.abs(b).max()
            d = np.abs(a - b)
            # iteration 0 differs by re-association only where no activation sits on a kink; the few PReLU / max-pool
            # elements within ~1e-7 of theirs take the other branch in one of the two runs (the sync-BN statistics are summed
            # in another order), which moves single gradient entries by ~1e-4 of the scale.  Measured on G at iteration 0:
            # below 2e-4 / 2e-6 (max / mean, the bounds this test had) with the register-staged GEMM kernels, 2.55e-4 / 3.34e-6
            # with the LDS-direct-load kernels (another k order, other elements flip); each run on its own matches the oracle
            # to rel-l2 1e-6 at batch 8 (test_training_steps_gradients_and_adam_state, profiles/r02b_step_gradients_vs_oracle.txt).
            assert d.max() <= (6e-4 if it == 0 else 5e-2) * scale and d.mean() <= (8e-6 if it == 0 else 1e-3) * scale, \
                (it, name, d.max() / scale, d.mean() / scale)
    for a, b, name in ((r0[2], full[2], "G"), (r0[3], full[3], "D")):
        # Adam's first steps move every weight by ~lr * sign(g) (lr = 1e-3): a weight whose gradient is ~0 may end 2 lr apart
        # per iteration, 4 lr after the two iterations (measured max: below 2.5e-3 with the register-staged kernels, 2.73e-3 with
        # the LDS-direct-load kernels; mean 1.8e-5)
        d = np.abs(a - b)
        assert d.max() <= 4.2e-3 and d.mean() <= 3e-5, (name, d.max(), d.mean())


def _strict_worker(rank, world, port, q):
    try:
        sys.path.insert(0, ROOT)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                          LOCAL_WORLD_SIZE=str(world), CG_COMM="abi", CG_COMM_STRICT="1", CATGAN_DIST_BACKEND="gloo")
        cg = importlib.import_module("cat-generator_amd")
        try:
            cg.parallel.init_from_env()
            q.put((rank, "no error", cg.parallel.comm_info()))
        except RuntimeError as e:
            q.put((rank, str(e), cg.parallel.comm_info()))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:
        q.put((rank, f"unexpected {type(e).__name__}: {e}", None))


@pytest.mark.timeout(300)
def test_strict_transport_refuses_the_fallback_on_every_rank():
    """CG_COMM_STRICT=1 (bench.py --strict-comm): where the default would move every rank to torch.distributed - here: two ranks
    sharing the box's one GPU, which RCCL refuses - init raises instead, on EVERY rank (the decision is collective), so a multi-GPU
    number measured in strict mode can only be cg_comm_*'s.  comm_info() names the transport and the reason either way."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29950 + os.getpid() % 40
    procs = [ctx.Process(target=_strict_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=200) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(60)
    for rank, msg, info in res:
        assert "CG_COMM_STRICT=1" in msg and "refusing to fall back" in msg, (rank, msg)
        assert info["backend"] == "torch" and info["strict"] is True and "not the engine's transport" in info["transport"]
        assert "fallback_reason" in info
