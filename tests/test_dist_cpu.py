"""CPU tests (-m "not gpu"): the N > 1 path with gloo, world size 2 -- batch sharding and
the bucketed gradient all-reduce of the training step."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dms_amd.dist import allreduce_gradients, shard_range


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
