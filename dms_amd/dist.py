# -*- coding: utf-8 -*-
"""Multi-GPU helpers: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

The hot path shards over the batch of (image, query) pairs (each sample's forward is
independent; the reference is literally batch-1, dmn.py:142), so inference needs NO collective:
``shard_range`` hands every rank a contiguous slice.  The only exchange step is the gradient
all-reduce of the training step (BASELINE configs[3]); ``allreduce_gradients`` does it in a few
large flat buckets -- xGMI is point-to-point, so few big messages beat many small ones.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise the default process group from torchrun's environment (no-op if WORLD_SIZE<=1).
    Returns (rank, local_rank, world_size)."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_range(n_items, rank, world):
    """Contiguous [lo, hi) slice of n_items for `rank`; sizes differ by at most one."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world: %d/%d" % (rank, world))
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def allreduce_gradients(params, bucket_mb=256, average=True, group=None):
    """Sum (or average) the .grad of `params` across ranks with flat fp32 buckets.
    Returns the number of all-reduce calls issued.  No-op without a process group."""
    if not dist.is_available() or not dist.is_initialized():
        return 0
    world = dist.get_world_size(group)
    if world == 1:
        return 0
    grads = [p.grad for p in params if p.grad is not None]
    limit = int(bucket_mb * (1 << 20) // 4)
    calls, bucket, count = 0, [], 0

    def flush():
        nonlocal calls, bucket, count
        if not bucket:
            return
        flat = torch.cat([g.reshape(-1) for g in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        if average:
            flat.div_(world)
        off = 0
        for g in bucket:
            n = g.numel()
            g.copy_(flat[off:off + n].view_as(g))
            off += n
        calls += 1
        bucket, count = [], 0

    for g in grads:
        if count and count + g.numel() > limit:
            flush()
        bucket.append(g)
        count += g.numel()
    flush()
    return calls
