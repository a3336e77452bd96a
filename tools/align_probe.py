"""Does 1 KiB alignment of the per-wave U segments matter?  d = 1005 vs 1024 at T=5120, B=32."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dms_amd import ops

dev = "cuda"


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


T, B = 5120, 32
for d in (1005, 1024):
    for k in (4, 3, 2):
        u = torch.randn(T, B, d, k, device=dev)
        x = torch.randn(T, B, d, device=dev) if k == 3 else None
        bias = torch.randn(2 * d, device=dev) * 0.1
        if k == 2:
            ms = timeit(lambda: ops.sru_state(u, bias))
            nb = 8 * T * B * d
        else:
            ms = timeit(lambda: ops.sru_forward_raw(u, x, bias, None, "tanh", 1.0, nchunks=1))
            nb = 20 * T * B * d
        print("d=%d k=%d: %.3f ms  %.0f GB/s (%.1f%%)   useful-bytes(d=1005) frac %.1f%%"
              % (d, k, ms, nb / ms / 1e6, nb / ms / 1e6 / 80, nb * 1005 / d / ms / 1e6 / 80), flush=True)
        del u, x
