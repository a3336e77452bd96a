"""Search rocBLAS + hipBLASLt solutions for the fp32 GEMM shapes of the 320 px / B = 64 TRAINING step
(forward, dgrad and wgrad of the hot path's linear maps) with PyTorch's TunableOp, write the table, then
time the step with and without it.

    python tools/tune_train_gemms.py [out.csv] [batch] [max_ms_per_solution] [max_iters]

Progress: TunableOp's verbose log goes to stderr (redirect it to a file under gpurun_out/)."""
import os
import sys
import time

os.environ.setdefault("PYTORCH_TUNABLEOP_VERBOSE", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.cuda.tunable as tunable
import torch.nn.functional as F

import bench

out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/tunableop_train320.csv"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
max_ms = int(sys.argv[3]) if len(sys.argv) > 3 else 30
max_it = int(sys.argv[4]) if len(sys.argv) > 4 else 5
dev = "cuda:0"
net = bench.build_model(dev, high_res=False).train()
g = torch.Generator(device=dev).manual_seed(bench.SEED)
img = torch.randn((B, 3, 320, 320), device=dev, generator=g)
words = torch.randint(0, 1000, (B, 20), device=dev, generator=g)
mask = (torch.rand((B, 1, 320, 320), device=dev, generator=g) > 0.5).float()
crit = torch.nn.BCEWithLogitsLoss()


def step():
    net.zero_grad(set_to_none=True)
    o = F.interpolate(net(img, words), size=mask.shape[-2:], mode="bilinear", align_corners=False)
    loss = crit(o, mask)
    loss.backward()
    return loss


def timeit(n=5):
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        l = step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, float(l)


base_ms, l0 = timeit()
print("untuned: %.2f ms/step (forward + backward, no optimizer), loss %.6f" % (base_ms, l0), flush=True)
tunable.enable(True)
tunable.tuning_enable(True)
tunable.set_max_tuning_duration(max_ms)
tunable.set_max_tuning_iterations(max_it)
tunable.set_filename(out)
t0 = time.perf_counter()
step()
torch.cuda.synchronize()
print("tuning pass: %.1f s, %d entries" % (time.perf_counter() - t0, len(tunable.get_results())), flush=True)
tunable.tuning_enable(False)          # (TunableOp writes the table to `out` itself)
tuned_ms, l1 = timeit()
print("tuned: %.2f ms/step (untuned %.2f); loss %.6f vs %.6f" % (tuned_ms, base_ms, l1, l0), flush=True)
