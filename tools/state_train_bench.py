#!/usr/bin/env python
"""State-only training scans (k = 2: dmn_sru_state_fwd_train / dmn_sru_state_bwd) at the configs[3] size over ring depths."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dms_amd import _lib
dev = "cuda"
T, B, d = 1900, 64, 1005
g = torch.Generator(device=dev).manual_seed(1)
u2 = torch.randn((T, B, d, 2), device=dev, generator=g)
bias = 0.1 * torch.randn((2 * d,), device=dev, generator=g)
c_all = torch.empty((T, B, d), device=dev); c_last = torch.empty((B, d), device=dev)
gl = torch.randn((B, d), device=dev, generator=g)
gu = torch.empty_like(u2); gb = torch.empty((d,), device=dev); gc0 = torch.empty((B, d), device=dev)
st = torch.cuda.current_stream().cuda_stream
P = lambda t: t.data_ptr()


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for depth in (4, 6, 8, 16, 24, 32):
    os.environ["DMN_SRU_DEPTH"] = str(depth)
    ms = timeit(lambda: _lib.call("dmn_sru_state_fwd_train", P(u2), P(bias), None, P(c_all), P(c_last), T, B, d, st))
    print("state fwd +c depth %2d: %.4f ms  %.0f GB/s (%.1f %%)" % (depth, ms, 12.0 * T * B * d / ms / 1e6, 12.0 * T * B * d / ms / 8e7), flush=True)
os.environ.pop("DMN_SRU_DEPTH")
for depth in (3, 4, 6, 8, 16):
    os.environ["DMN_SRU_BWD_DEPTH"] = str(depth)
    ms = timeit(lambda: _lib.call("dmn_sru_state_bwd", P(u2), P(bias), None, P(c_all), P(gl), P(gu), P(gb), P(gc0), None, T, B, d, st))
    print("state bwd    depth %2d: %.4f ms  %.0f GB/s (%.1f %%)" % (depth, ms, 20.0 * T * B * d / ms / 1e6, 20.0 * T * B * d / ms / 8e7), flush=True)
