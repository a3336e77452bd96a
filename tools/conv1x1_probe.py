"""Probe: DPN92 1x1 convolutions at B=32 / 512 px: MIOpen conv2d vs a batched library GEMM
W (Cout x Cin) . X[n] (Cin x HW) on the same NCHW tensors; plus a device fill for the
write-only bandwidth ceiling."""
import sys
import time

import torch
import torch.nn.functional as F

dev = "cuda:0"
B = 32


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


shapes = [("c2 1x1a", 336, 96, 128), ("c2 1x1c", 96, 272, 128), ("c2 proj", 64, 288, 128),
          ("c3 1x1a", 704, 192, 64), ("c3 1x1c", 192, 544, 64),
          ("c4 1x1a", 1552, 384, 32), ("c4 1x1c", 384, 1048, 32),
          ("c5 1x1a", 2560, 768, 16), ("c5 1x1c", 768, 2176, 16)]
for backend in ("cublas", "cublaslt"):
    torch.backends.cuda.preferred_blas_library(backend)
    for name, cin, cout, H in shapes:
        x = torch.randn(B, cin, H, H, device=dev)
        w = torch.randn(cout, cin, 1, 1, device=dev) * 0.05
        w2 = w.view(cout, cin)
        with torch.no_grad():
            t_conv = timeit(lambda: F.conv2d(x, w))
            t_mm = timeit(lambda: torch.matmul(w2, x.view(B, cin, H * H)))
            err = ((torch.matmul(w2, x.view(B, cin, H * H)).view(B, cout, H, H) - F.conv2d(x, w)).abs().max()).item()
        fl = 2.0 * B * H * H * cin * cout
        print("%-8s %-8s conv %.3f ms (%.0f TF)  bmm %.3f ms (%.0f TF)  maxabs diff %.1e"
              % (backend, name, t_conv, fl / t_conv / 1e9, t_mm, fl / t_mm / 1e9, err), flush=True)
buf = torch.empty(2_600_000_000 // 4, device=dev)
t = timeit(lambda: buf.fill_(1.0))
print("fill 2.6 GB: %.3f ms  %.0f GB/s" % (t, 2.6e9 / t / 1e6), flush=True)
src = torch.empty_like(buf)
t = timeit(lambda: buf.copy_(src))
print("copy 2.6 GB: %.3f ms  %.0f GB/s (r+w)" % (t, 5.2e9 / t / 1e6), flush=True)
