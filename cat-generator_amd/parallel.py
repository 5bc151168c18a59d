"""Data-parallel plumbing: one process per GPU, torch.distributed over RCCL/xGMI ("nccl") on the GPU box,
gloo on CPU for the world_size-2 tests.  The reference has no multi-GPU code; the exchange
steps are the ones §8e derives:
  * all-reduce(mean) of the flat gradient vector of a net, BEFORE penalty/clamp/Adam (adversarial.lua:92-112 order)
  * sync-BN: all-reduce(sum) of the 2C fp64 batch statistics (forward) and of (sum dy, sum dy*xhat) (backward)
"""
import os

import torch
import torch.distributed as dist

_S = {"world": 1, "rank": 0, "init": False, "sync_bn": True}


def init_from_env(backend=None):
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT (torch.distributed.run contract)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available():
        torch.cuda.set_device(local % torch.cuda.device_count())
    if world > 1 and not _S["init"]:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = backend or os.environ.get("CATGAN_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
        _S["init"] = True
    _S["world"], _S["rank"] = world, rank
    return rank, world


def attach(world, rank):
    """Use an already initialised process group (tests)."""
    _S["world"], _S["rank"] = world, rank


def shutdown():
    if _S["init"]:
        dist.destroy_process_group()
        _S["init"] = False
    _S["world"], _S["rank"] = 1, 0


def world_size():
    return _S["world"]


def rank():
    return _S["rank"]


def set_sync_bn(flag):
    _S["sync_bn"] = bool(flag)


def sync_bn_active():
    return _S["world"] > 1 and _S["sync_bn"]


def allreduce_sum_(t):
    """In-place SUM over ranks of a torch tensor (fp64 statistics)."""
    if _S["world"] > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def allreduce_mean_(t):
    """In-place MEAN over ranks of a flat fp32 gradient vector (torch tensor)."""
    if _S["world"] > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        if t.is_cuda:
            from .tensor import lib, stream
            lib().scale(stream(), t.data_ptr(), 1.0 / _S["world"], t.numel())
        else:
            t.mul_(1.0 / _S["world"])
    return t


class _Pending:
    """An all-reduce(mean) in flight on RCCL's stream; finish() makes the compute stream wait for it and applies
    the 1/world scale."""

    def __init__(self, t, work):
        self.t, self.work = t, work

    def finish(self):
        if self.work is not None:
            self.work.wait()
            if self.t.is_cuda:
                from .tensor import lib, stream
                lib().scale(stream(), self.t.data_ptr(), 1.0 / _S["world"], self.t.numel())
            else:
                self.t.mul_(1.0 / _S["world"])
        return self.t


def allreduce_mean_async(t):
    """Start the gradient all-reduce without blocking the compute stream (xGMI transfer overlaps the kernels
    launched until finish())."""
    if _S["world"] > 1:
        return _Pending(t, dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True))
    return _Pending(t, None)


def allreduce_sum_host(values):
    """SUM over ranks of a few host scalars (accuracy gate, confusion counts): returns a list of floats."""
    if _S["world"] <= 1:
        return [float(v) for v in values]
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(v) for v in t.cpu().tolist()]


def barrier():
    if _S["world"] > 1:
        if torch.cuda.is_available() and dist.get_backend() == "nccl":
            dist.barrier(device_ids=[torch.cuda.current_device()])   # explicit device: no guessing from the rank
        else:
            dist.barrier()
