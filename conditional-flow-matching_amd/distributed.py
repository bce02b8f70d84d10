"""Multi-GPU sharding of the hot path: one process per GPU, no collective inside the path.

The reference's DDP script solves OT on each rank's local batch only
(examples/images/cifar10/train_cifar10_ddp.py:74,92,167-169); the only exchange the north
star names is one all-gather of the final samples over xGMI (RCCL).  ``backend="nccl"`` is
RCCL on ROCm; ``gloo`` is used by the CPU tests.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from torchrun's env (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            # CFM_DIST_BACKEND=gloo lets the multi-rank path be exercised on a single-GPU box
            backend = os.environ.get("CFM_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if torch.cuda.is_available():
            torch.cuda.set_device(local % torch.cuda.device_count())
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    elif torch.cuda.is_available():
        torch.cuda.set_device(local % torch.cuda.device_count())
    return rank, local, world


def shard_seed(base, rank):
    """Each rank draws its own minibatch: seed 1000 + rank style."""
    return int(base) + int(rank)


def all_gather_samples(x, direct=None, force=False):
    """[B,d] per rank -> [world*B, d] on every rank (identity at world=1).

    ``force=True``: with an initialised process group of ONE rank the collective (or, ``direct``, one send + one
    receive to the rank itself) is issued anyway instead of returning ``x`` — the N > 1 code on the real backend of a
    one-GPU box (tests/test_gpu_rccl_world1.py).

    Default: one RCCL all-gather.  ``direct=True`` (or CFM_ALLGATHER=direct): the fully connected exchange SURVEY
    8(e) describes — every rank posts world-1 sends of its block and world-1 receives into the output in one
    batch, so each of the 7 xGMI links of an MI355X carries exactly one block each way instead of a ring relaying
    every block over one link.  Which one is faster on an 8-GPU node is not measured here (one GPU per box);
    both give the same bytes (tests/test_distributed_gloo.py runs both over gloo)."""
    if not dist.is_available() or not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return x
    world, rank = dist.get_world_size(), dist.get_rank()
    x = x.contiguous()
    out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    if direct is None:
        direct = os.environ.get("CFM_ALLGATHER", "") == "direct"
    if not direct:
        dist.all_gather_into_tensor(out, x)
        return out
    B = x.shape[0]
    if world == 1:                                 # (forced) the exchange with the only peer there is: this rank
        ops = [dist.P2POp(dist.isend, x, rank), dist.P2POp(dist.irecv, out[:B], rank)]
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        return out
    out[rank * B:(rank + 1) * B].copy_(x)
    ops = []
    for k in range(1, world):                      # peer order staggered by rank: no two ranks start on the same link
        peer = (rank + k) % world
        src = (rank - k) % world
        ops.append(dist.P2POp(dist.isend, x, peer))
        ops.append(dist.P2POp(dist.irecv, out[src * B:(src + 1) * B], src))
    for req in dist.batch_isend_irecv(ops):
        req.wait()
    return out


def max_over_ranks(value, device=None):
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    dev = device or (torch.device("cuda", torch.cuda.current_device())
                     if dist.get_backend() == "nccl" else torch.device("cpu"))
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
