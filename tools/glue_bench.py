#!/usr/bin/env python
"""Microbench of the glue kernels at the 512px / B=32 shapes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dms_amd import ops

dev = "cuda"


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
for (Cx, Cf, H) in [(1005, 2688, 16), (512, 1552, 32), (256, 704, 64), (128, 336, 128), (64, 64, 256)]:
    x = torch.randn(B, Cx, H, H, device=dev)
    f = torch.randn(B, Cf, H, H, device=dev)
    ms = timeit(lambda: ops.cat_upsample2x(x, f))
    nb = 4 * B * (Cx + Cf) * H * H * 5
    print("cat_upsample2x %4d+%4d @%3d : %7.3f ms %7.1f GB/s (%.1f%%)" % (Cx, Cf, H, ms, nb / ms / 1e6, nb / ms / 1e6 / 80), flush=True)
    del x, f
L, K, C8, S, M = 20, 10, 2696, 256, 1000
A = torch.randn(B, S, M, device=dev); Bt = torch.randn(L, B, M, device=dev)
P = torch.randn(L, B, K, S, device=dev); W = torch.randn(M, 4706, device=dev)[:, 4696:]
ms = timeit(lambda: ops.mix_bcast_add(A, Bt, P, W))
nb = 4 * B * (S * M + L * M + L * K * S + S * L * M)
print("mix_bcast_add : %7.3f ms %7.1f GB/s (%.1f%%)" % (ms, nb / ms / 1e6, nb / ms / 1e6 / 80))
V = torch.randn(B, C8, S, device=dev); Fi = torch.rand(L, B, K, C8, device=dev)
ms = timeit(lambda: ops.dynfilter(V, Fi))
fl = 2.0 * B * L * K * C8 * S
print("dynfilter_fwd : %7.3f ms %7.1f GFLOP/s (%.1f%% of 157 TF) %7.1f GB/s" % (ms, fl / ms / 1e6, fl / ms / 1e6 / 1570, 4 * B * (C8 * S + L * K * C8 + L * K * S) / ms / 1e6))
vis = torch.randn(B, 2688, 16, 16, device=dev)
ms = timeit(lambda: ops.cat_coords(vis))
print("cat_coords : %7.3f ms %7.1f GB/s" % (ms, 4 * B * 256 * (2 * 2688 + 8) / ms / 1e6))
