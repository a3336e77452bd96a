# -*- coding: utf-8 -*-
"""Multi-GPU helpers: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

The hot path shards over the batch of (image, query) pairs (each sample's forward is
independent; the reference is literally batch-1, dmn.py:142), so inference needs NO collective:
``shard_range`` hands every rank a contiguous slice.  The only exchange step is the gradient
all-reduce of the training step (BASELINE configs[3]): ``OverlappedGradReducer`` keeps every
``.grad`` as a view into a few large flat buckets and launches each bucket's RCCL all-reduce from
an autograd hook as soon as its last gradient has been written, so the exchange overlaps the rest
of backward (``allreduce_gradients`` is the simple after-backward form).
"""
import os
import sys

import torch
import torch.distributed as dist


_DEBUG = bool(os.environ.get("DMN_REDUCER_DEBUG"))     # trace finish() bucket by bucket on stderr


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


class OverlappedGradReducer(object):
    """Gradient all-reduce of the training step (BASELINE configs[3]), overlapped with backward.

    * Every parameter's ``.grad`` is a VIEW into one of a few pre-allocated flat fp32 buckets
      (``bucket_mb`` each; parameters in reverse registration order, roughly the order in which
      backward produces their gradients), so a bucket is reduced in place: no ``torch.cat``, no
      copy back -- the 455 MB of gradients are touched by RCCL only.
    * A post-accumulate hook per parameter counts the bucket's gradients as they arrive; the
      bucket's ``all_reduce(async_op=True)`` is launched from the hook of its LAST gradient, i.e.
      while backward is still running on the layers in front of it (RCCL runs on its own stream;
      xGMI is point-to-point, so few large buckets beat many small ones).
    * ``finish()`` (after ``backward()``, before ``optimizer.step()``) launches what did not
      fill up (parameters without a gradient this step), waits for all of it and averages.
    * Collectives go out strictly in bucket-index order on every rank (a bucket that fills up
      early waits for the buckets in front of it), so ranks whose graphs produce gradients in a
      different order -- or leave different parameters without one -- still issue matching
      collectives.  Parameters that received no gradient in the FIRST step are taken to be unused
      for good (their bucket no longer waits for them: it would otherwise hold back every bucket
      behind it until ``finish()``); if one of them does get a gradient later, after its bucket has
      gone out, that raises.  ONE ``backward()`` per ``finish()``: a gradient that arrives for a bucket
      whose exchange is already in flight (a second backward, ``retain_graph`` replays,
      gradient accumulation) raises instead of being reduced half-way.

    Use ``reducer.zero_grad()`` instead of ``optimizer.zero_grad(set_to_none=True)``, which would
    drop the views.  Without a process group (or world size 1) everything degrades to plain
    local gradients.  ``launch_log`` records (bucket index, launched_during_backward) per step.
    With backend gloo and device buckets the exchange is staged through pinned host memory (see
    ``_staged`` below): the copies overlap backward, the host all-reduce runs in ``finish()``.
    """

    def __init__(self, params, bucket_mb=128, average=True, group=None, collective_at_world_1=False):
        self.params = [p for p in params if p.requires_grad]
        self.group, self.average = group, average
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        # test hook: issue the collectives even in a one-rank group (exercises the RCCL path on one GPU)
        self._collective = self.world > 1 or (collective_at_world_1 and dist.is_available() and dist.is_initialized())
        limit = max(int(bucket_mb * (1 << 20) // 4), 1)
        # buckets over the parameters in reverse order
        plan, cur, count = [], [], 0
        for p in reversed(self.params):
            if p.dtype != torch.float32:
                raise TypeError("OverlappedGradReducer: fp32 parameters only")
            if cur and count + p.numel() > limit:
                plan.append(cur)
                cur, count = [], 0
            cur.append(p)
            count += p.numel()
        if cur:
            plan.append(cur)
        self.buckets, self._bucket_of, self._expected = [], {}, []
        for bi, ps in enumerate(plan):
            flat = torch.zeros(sum(p.numel() for p in ps), dtype=torch.float32, device=ps[0].device)
            off = 0
            for p in ps:
                p.grad = flat[off:off + p.numel()].view_as(p)
                off += p.numel()
                self._bucket_of[p] = bi
            self.buckets.append(flat)
            self._expected.append(len(ps))
        self._arrived = [0] * len(self.buckets)
        self._work = [None] * len(self.buckets)
        self._next = 0                                   # first bucket whose collective has not been issued
        self._learned = False                            # _expected still counts every parameter of a bucket
        self._in_backward = True
        self.launch_log = []
        # backend gloo + buckets in device memory (the one-GPU rehearsal of the N > 1 path): the exchange
        # is staged through pinned host memory by the reducer itself.  ProcessGroupGloo's own device-tensor
        # path does not return from its SECOND 128 MB all-reduce on this stack -- reproduced without this
        # class, without hooks, from the main thread and even in a ONE-rank group (tools/gloo_hook_repro.py,
        # profiles/r3_gloo_device_allreduce_repro.md: its two worker threads spin inside the HIP runtime;
        # 1 MB tensors and the same stream / pinned-copy / worker-thread pattern in plain torch are fine) --
        # so the cause is in that backend, not in how the collectives are issued here.  Host tensors are
        # what gloo reduces anyway.  RCCL ("nccl") reduces the device buckets in place.
        self._staged = (self._collective and self.buckets and self.buckets[0].is_cuda
                        and dist.get_backend(group) == "gloo")
        if self._staged:
            self._side = torch.cuda.Stream(device=self.buckets[0].device)
            self._host = [torch.empty(b.numel(), dtype=b.dtype, pin_memory=True) for b in self.buckets]
        self._handles = [p.register_post_accumulate_grad_hook(self._hook) for p in self.params]

    def _launch(self, bi):
        if self._work[bi] is not None:
            return
        self.launch_log.append((bi, self._in_backward))
        if not self._collective:
            self._work[bi] = True
        elif self._staged:
            # gloo with device buckets: start the bucket's copy to pinned host memory on the side
            # stream now (it overlaps the rest of backward); the host all-reduce runs in finish()
            ev = torch.cuda.Event()
            ev.record()
            with torch.cuda.stream(self._side):
                self._side.wait_event(ev)
                self._host[bi].copy_(self.buckets[bi], non_blocking=True)
                done = torch.cuda.Event()
                done.record(self._side)
            self._work[bi] = done
        else:
            self._work[bi] = dist.all_reduce(self.buckets[bi], op=dist.ReduceOp.SUM, group=self.group,
                                             async_op=True)

    def _hook(self, p):
        bi = self._bucket_of[p]
        if p.grad.data_ptr() < self.buckets[bi].data_ptr() or p.grad.numel() != p.numel():
            raise RuntimeError("OverlappedGradReducer: a .grad no longer views its bucket "
                               "(use reducer.zero_grad(), not zero_grad(set_to_none=True))")
        if self._work[bi] is not None:
            raise RuntimeError("OverlappedGradReducer: a gradient arrived for bucket %d after its exchange was "
                               "launched -- one backward() per finish() (no gradient accumulation or "
                               "retain_graph replays between them), and the set of parameters that receive "
                               "a gradient must not grow after the first step" % bi)
        self._arrived[bi] += 1
        # strictly in bucket order: a full bucket goes out once every bucket in front of it has
        while self._next < len(self.buckets) and self._arrived[self._next] == self._expected[self._next]:
            self._launch(self._next)
            self._next += 1

    def zero_grad(self):
        for flat in self.buckets:
            flat.zero_()

    def finish(self):
        """Wait for every bucket (launching those that never filled up) and average.
        Returns the number of all-reduce calls of this step."""
        self._in_backward = False
        for bi in range(self._next, len(self.buckets)):
            self._launch(bi)
        self._next = len(self.buckets)
        if self._staged:
            works = []
            for bi, done in enumerate(self._work):
                done.synchronize()                                   # the bucket is in host memory
                works.append(dist.all_reduce(self._host[bi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            for bi, w in enumerate(works):
                if _DEBUG:
                    print("[reducer] rank %d waiting for host bucket %d of %d" % (dist.get_rank(self.group), bi, len(works)),
                          file=sys.stderr, flush=True)
                w.wait()
                self.buckets[bi].copy_(self._host[bi], non_blocking=True)
                if self.average:
                    self.buckets[bi].div_(self.world)
        else:
            for bi, w in enumerate(self._work):
                if w is not True:
                    if _DEBUG:
                        print("[reducer] rank %d waiting for bucket %d of %d (launch order %s)" % (
                            dist.get_rank(self.group), bi, len(self._work), [b for b, _ in self.launch_log[-len(self._work):]]),
                            file=sys.stderr, flush=True)
                    w.wait()
                if self.average and self.world > 1:
                    self.buckets[bi].div_(self.world)
        calls = len(self._work)
        if not self._learned:
            # static graph assumed from here on: what did not arrive in the first step is not waited for
            self._expected = list(self._arrived)
            self._learned = True
        self._arrived = [0] * len(self.buckets)
        self._work = [None] * len(self.buckets)
        self._next = 0
        self._in_backward = True
        return calls

    def remove(self):
        for h in self._handles:
            h.remove()
        self._handles = []


def allreduce_gradients(params, bucket_mb=256, average=True, group=None):
    """Sum (or average) the .grad of `params` across ranks AFTER backward, in few large calls.
    Gradients that already live in one flat buffer (OverlappedGradReducer's views) are not
    copied; for ad-hoc use on plain .grad tensors they are flattened per bucket.  Prefer
    OverlappedGradReducer for the training loop.  Returns the number of all-reduce calls."""
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
        flat = torch._utils._flatten_dense_tensors(bucket)
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        if average:
            flat.div_(world)
        for g, r in zip(bucket, torch._utils._unflatten_dense_tensors(flat, bucket)):
            g.copy_(r)
        calls += 1
        bucket, count = [], 0

    for g in grads:
        if count and count + g.numel() > limit:
            flush()
        bucket.append(g)
        count += g.numel()
    flush()
    return calls
