"""Search rocBLAS + hipBLASLt solutions for every fp32 GEMM shape of the B=32 / 512 px inference
step with PyTorch's TunableOp, write the table, then time the step with and without it.

    python tools/tune_gemms.py [out.csv] [batch] [max_ms_per_solution] [max_iters]

Progress: TunableOp's verbose log goes to stderr (redirect it to a file under gpurun_out/)."""
import os
import sys
import time

os.environ.setdefault("PYTORCH_TUNABLEOP_VERBOSE", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.cuda.tunable as tunable

import bench

out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/tunableop_b32_512.csv"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
max_ms = int(sys.argv[3]) if len(sys.argv) > 3 else 15
max_it = int(sys.argv[4]) if len(sys.argv) > 4 else 10
dev = "cuda:0"
net = bench.build_model(dev)
g = torch.Generator(device=dev).manual_seed(bench.SEED)
img = torch.randn((B, 3, 512, 512), device=dev, generator=g)
words = torch.randint(0, 1000, (B, 20), device=dev, generator=g)


def timeit(n=10):
    with torch.no_grad():
        for _ in range(3):
            y = net(img, words)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            y = net(img, words)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, y


base_ms, y0 = timeit()
print("untuned: %.2f ms/step" % base_ms, flush=True)

tunable.enable(True)
tunable.tuning_enable(True)
tunable.set_max_tuning_duration(max_ms)
tunable.set_max_tuning_iterations(max_it)
tunable.set_filename(out)
t0 = time.perf_counter()
# TunableOp's tuning mode cannot duplicate GEMM outputs that are row-slice views (the in-place DPN
# stages): tune on the per-block merge path, whose GEMMs have the same m, n, k, ld keys
net.langvis.base.stage_inplace = False
with torch.no_grad():
    net(img, words)
    torch.cuda.synchronize()
net.langvis.base.stage_inplace = True
print("tuning pass: %.1f s, %d entries" % (time.perf_counter() - t0, len(tunable.get_results())), flush=True)
tunable.tuning_enable(False)
tuned_ms, y1 = timeit()
err = ((y1 - y0).abs().max() / y0.abs().max()).item()
print("tuned: %.2f ms/step (untuned %.2f); rel diff of the output %.2e" % (tuned_ms, base_ms, err), flush=True)
