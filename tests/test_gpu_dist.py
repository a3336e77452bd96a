"""GPU test (-m gpu): the N > 1 training exchange with DEVICE buckets, world size 2 on one card.
RCCL refuses two ranks on one device, so the backend is gloo and OverlappedGradReducer stages the
buckets through pinned host memory (dms_amd/dist.py, `_staged`) -- the path bench.py's one-GPU
rehearsal of --workload train320 takes (profiles/r2_train320_2rank_gloo.md)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dms_amd.dist import OverlappedGradReducer, shard_range

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _toy():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(64, 300), torch.nn.Tanh(), torch.nn.Linear(300, 200), torch.nn.Tanh(),
                               torch.nn.Linear(200, 1))


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda", 0)
        model = _toy().to(dev)
        red = OverlappedGradReducer(list(model.parameters()), bucket_mb=0.1, average=True)
        assert red._staged and len(red.buckets) >= 3
        x = (torch.arange(16 * 64, dtype=torch.float32).view(16, 64) % 13 - 6.0) / 10.0
        lo, hi = shard_range(16, rank, world)
        steps = []
        for step in range(3):                                     # the gloo device path hung on the SECOND step
            red.zero_grad()
            red.launch_log = []
            loss = model(x[lo:hi].to(dev) * (step + 1)).pow(2).sum() / (hi - lo)
            loss.backward()
            in_bwd = list(red.launch_log)
            calls = red.finish()
            torch.cuda.synchronize()
            steps.append({"grad": torch.cat([p.grad.reshape(-1) for p in model.parameters()]).cpu(),
                          "calls": calls, "in_bwd": in_bwd})
        if rank == 0:
            torch.save(steps, out)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(240)
def test_device_buckets_over_gloo_world2_three_steps_equal_the_single_process_gradient(tmp_path):
    assert torch.cuda.is_available(), "this test needs a real MI355X"
    out = str(tmp_path / "g.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    steps = torch.load(out)
    x = (torch.arange(16 * 64, dtype=torch.float32).view(16, 64) % 13 - 6.0) / 10.0
    for step, got in enumerate(steps):
        model = _toy()
        model(x * (step + 1)).pow(2).mean().backward()
        ref = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
        torch.testing.assert_close(got["grad"], ref, rtol=2e-5, atol=1e-6)
        assert got["calls"] == len(got["in_bwd"]) >= 3
        assert all(inb for _, inb in got["in_bwd"])               # every bucket's copy started from a hook


def _rccl_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world)       # "nccl" IS RCCL on ROCm
    try:
        dev = torch.device("cuda", 0)
        model = _toy().to(dev)
        red = OverlappedGradReducer(list(model.parameters()), bucket_mb=0.1, average=True, collective_at_world_1=True)
        assert not red._staged and len(red.buckets) >= 3
        x = (torch.arange(16 * 64, dtype=torch.float32).view(16, 64) % 13 - 6.0) / 10.0
        steps = []
        for step in range(3):
            red.zero_grad()
            red.launch_log = []
            model(x.to(dev) * (step + 1)).pow(2).mean().backward()
            in_bwd = list(red.launch_log)
            calls = red.finish()
            torch.cuda.synchronize()
            steps.append({"grad": torch.cat([p.grad.reshape(-1) for p in model.parameters()]).cpu(),
                          "calls": calls, "in_bwd": in_bwd})
        t = torch.ones(4, device=dev)
        dist.all_reduce(t)
        dist.barrier()
        torch.save({"steps": steps, "sum": t.cpu()}, out)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(240)
def test_device_buckets_through_rccl_in_place_one_rank_group(tmp_path):
    """The RCCL code path of the reducer (async all-reduce of the device buckets IN PLACE, launched from
    autograd hooks, waited for in finish()) on the one GPU there is: a one-rank "nccl" group with the
    reducer told to issue its collectives anyway.  Gradients must equal plain autograd's."""
    assert torch.cuda.is_available(), "this test needs a real MI355X"
    out = str(tmp_path / "r.pt")
    mp.spawn(_rccl_worker, args=(1, _free_port(), out), nprocs=1, join=True)
    got = torch.load(out)
    assert torch.equal(got["sum"], torch.ones(4))
    x = (torch.arange(16 * 64, dtype=torch.float32).view(16, 64) % 13 - 6.0) / 10.0
    for step, g in enumerate(got["steps"]):
        model = _toy()
        model(x * (step + 1)).pow(2).mean().backward()
        ref = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
        torch.testing.assert_close(g["grad"], ref, rtol=2e-5, atol=1e-6)
        assert g["calls"] == len(g["in_bwd"]) >= 3 and all(inb for _, inb in g["in_bwd"])


@pytest.mark.timeout(420)
def test_bench_gpus_2_starts_its_own_ranks_and_reports_what_the_collective_saw():
    """`python bench.py --gpus 2` with no torchrun environment (the driver's form of the command): on this
    one-GPU box it must refuse; with the rehearsal hooks (both ranks on card 0, gloo instead of RCCL, which
    refuses two ranks on one device) the SAME launcher starts two ranks and the line's n_gpus /
    collective_ranks come from the process group, not from the flag."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "DMN_BENCH_DEVICE")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--batch", "2", "--steps", "2",
           "--warmup", "1", "--no-cpu-baseline", "--no-extras"]
    if torch.cuda.device_count() < 2:
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        assert r.returncode != 0 and b"refusing" in r.stderr
    env.update(DMN_BENCH_DEVICE="0", DMN_BENCH_BACKEND="gloo")
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=360)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-2000:]
    rows = [json.loads(ln) for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(rows) == 1                                            # rank 0 prints ONE line
    line = rows[0]
    assert line["n_gpus"] == 2 and line["collective_ranks"] == 2 and line["collective_backend"] == "gloo"
    assert line["config"]["global_batch"] == 4 and line["value"] > 0
