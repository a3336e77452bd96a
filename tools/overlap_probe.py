"""Does splitting the batch over HIP streams overlap the HBM-bound glue kernels of one part with
the MFMA-bound GEMMs of another?  B=32 on one stream vs 2 x 16 and 4 x 8 on separate streams."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

dev = "cuda:0"
B = 32
net = bench.build_model(dev)
g = torch.Generator(device=dev).manual_seed(bench.SEED)
img = torch.randn((B, 3, 512, 512), device=dev, generator=g)
words = torch.randint(0, 1000, (B, 20), device=dev, generator=g)


def run_split(parts, streams):
    cur = torch.cuda.current_stream()
    outs = []
    n = B // parts
    for p in range(parts):
        s = streams[p]
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            outs.append(net(img[p * n:(p + 1) * n], words[p * n:(p + 1) * n]))
    for s in streams[:parts]:
        cur.wait_stream(s)
    return torch.cat(outs)


def timeit(fn, n=10):
    with torch.no_grad():
        for _ in range(3):
            y = fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            y = fn()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, y


ms, y0 = timeit(lambda: net(img, words))
print("1 x 32 on one stream: %.2f ms" % ms, flush=True)
streams = [torch.cuda.Stream() for _ in range(4)]
for parts in (2, 4):
    ms, y = timeit(lambda: run_split(parts, streams))
    err = ((y - y0).abs().max() / y0.abs().max()).item()
    print("%d x %d on %d streams: %.2f ms (rel diff %.1e)" % (parts, B // parts, parts, ms, err), flush=True)
    same = [torch.cuda.current_stream()] * 4
    ms, y = timeit(lambda: run_split(parts, same))
    print("%d x %d on ONE stream:  %.2f ms" % (parts, B // parts, ms), flush=True)
