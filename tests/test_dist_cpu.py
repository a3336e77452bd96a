"""CPU tests (-m "not gpu"): the N > 1 path with gloo, world size 2 -- batch sharding and
the bucketed gradient all-reduce of the training step."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dms_amd.dist import OverlappedGradReducer, allreduce_gradients, shard_range


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 32, 256, 257):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 1))
        x = torch.arange(8 * 6, dtype=torch.float32).view(8, 6) / 10.0
        lo, hi = shard_range(8, rank, world)
        # loss = SUM over the global batch; averaging the per-rank grads of per-rank SUM/… is
        # checked against the single-process gradient of the mean loss
        loss = model(x[lo:hi]).pow(2).sum() / (hi - lo)
        loss.backward()
        calls = allreduce_gradients(list(model.parameters()), bucket_mb=0.00001, average=True)
        flat = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
        if rank == 0:
            torch.save({"grad": flat, "calls": calls}, out)
    finally:
        dist.destroy_process_group()


def test_bucketed_gradient_allreduce_world2(tmp_path):
    out = str(tmp_path / "g.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 1))
    x = torch.arange(8 * 6, dtype=torch.float32).view(8, 6) / 10.0
    model(x).pow(2).mean().backward()
    ref = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    assert got["calls"] >= 2                     # tiny bucket size forces several buckets
    torch.testing.assert_close(got["grad"], ref, rtol=1e-5, atol=1e-6)


def _toy():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 4), torch.nn.Tanh(),
                               torch.nn.Linear(4, 1))


def _overlap_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model = _toy()
        unused = torch.nn.Parameter(torch.ones(3))                  # never reaches the loss
        params = list(model.parameters()) + [unused]
        # 25 floats per bucket, reverse order: [unused 3 | L3.b 1 | L3.w 4 | L2.b 4] [L2.w 20 | L1.b 5] [L1.w 30]
        red = OverlappedGradReducer(params, bucket_mb=25 * 4 / (1 << 20), average=True)
        x = torch.arange(8 * 6, dtype=torch.float32).view(8, 6) / 10.0
        lo, hi = shard_range(8, rank, world)
        steps = []
        for step in range(2):                                       # second step: views survive zero_grad()
            red.zero_grad()
            red.launch_log = []
            seen = []
            hooks = [p.register_post_accumulate_grad_hook(lambda p, i=i: seen.append(i))
                     for i, p in enumerate(params)]
            loss = model(x[lo:hi] * (step + 1)).pow(2).sum() / (hi - lo)
            loss.backward()
            log_after_backward = list(red.launch_log)
            calls = red.finish()
            for h in hooks:
                h.remove()
            flat = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
            steps.append({"grad": flat.clone(), "calls": calls, "log_bwd": log_after_backward,
                          "log": list(red.launch_log), "seen": seen, "unused": unused.grad.clone(),
                          "views": all(p.grad.data_ptr() >= red.buckets[red._bucket_of[p]].data_ptr() for p in params)})
        if rank == 0:
            torch.save(steps, out)
    finally:
        dist.destroy_process_group()


def test_overlapped_gradient_allreduce_world2_launches_from_hooks_and_equals_single_process(tmp_path):
    """Buckets must be launched FROM the autograd hooks (during backward, last layers first), the
    bucket holding a gradient-less parameter by finish(); averaged gradients must equal the
    single-process gradient of the mean loss over the whole batch; .grad stay views of the buckets."""
    out = str(tmp_path / "g.pt")
    mp.spawn(_overlap_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    steps = torch.load(out)
    x = torch.arange(8 * 6, dtype=torch.float32).view(8, 6) / 10.0
    for step, got in enumerate(steps):
        model = _toy()
        model(x * (step + 1)).pow(2).mean().backward()
        ref = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
        torch.testing.assert_close(got["grad"], ref, rtol=1e-5, atol=1e-6)
        assert got["calls"] == 3 and got["views"]
        # collectives go out strictly in bucket order.  Step 0: bucket 0 holds the unused parameter, never
        # fills up and holds the others back until finish().  From step 1 on the reducer knows that
        # parameter receives no gradient: every bucket goes out during backward, last layers first.
        if step == 0:
            assert got["log_bwd"] == [] and got["log"] == [(0, False), (1, False), (2, False)]
        else:
            assert got["log_bwd"] == [(0, True), (1, True), (2, True)] and got["log"] == got["log_bwd"]
        # autograd's hook order: last layer's parameters first
        assert got["seen"][0] in (4, 5) and got["seen"][-1] in (0, 1) and 6 not in got["seen"]
        assert float(got["unused"].abs().sum()) == 0.0


def test_second_backward_before_finish_raises_instead_of_reducing_half_way():
    """A gradient that arrives for a bucket whose exchange is already in flight (second backward() before
    finish(): gradient accumulation, retain_graph) would be accumulated into a bucket RCCL is reducing:
    the reducer must raise.  After finish() the next step works again."""
    model = _toy()
    red = OverlappedGradReducer(list(model.parameters()), bucket_mb=25 * 4 / (1 << 20))
    x = torch.arange(8 * 6, dtype=torch.float32).view(8, 6) / 10.0
    red.zero_grad()
    model(x).pow(2).mean().backward()
    with pytest.raises(RuntimeError, match="one backward"):
        model(x).pow(2).mean().backward()
    red.finish()
    red.zero_grad()
    model(x).pow(2).mean().backward()
    assert red.finish() == len(red.buckets)
    ref = _toy()
    ref(x).pow(2).mean().backward()
    for p, q in zip(model.parameters(), ref.parameters()):
        torch.testing.assert_close(p.grad, q.grad)
    red.remove()
