#!/usr/bin/env python
"""dmn_sru_fwd_image (tail of the last mSRU layer, stored as the (B, Hm, H, W) map): plain vs time-chunked."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dms_amd import ops
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
for (T, B, d) in ((256, 32, 1005), (256, 64, 1005), (256, 128, 1005), (256, 8, 1005)):
    u = torch.randn((T, B, d, 3), device=dev, generator=g); x = torch.randn((T, B, d), device=dev, generator=g)
    bias = 0.1 * torch.randn((2 * d,), device=dev, generator=g); c0 = torch.randn((B, d), device=dev, generator=g)
    whole = torch.empty((B, d + 1008, T), device=dev)
    nb = 20 * T * B * d
    for nch in (1, 2, 4):
        fn = lambda: ops.sru_forward_image(u, x, bias, c0, "tanh", 1.0, whole[:, :d], nchunks=nch)
        for _ in range(3): fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): fn()
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 20
        print("T=%d B=%d nchunks=%d: %.4f ms %.0f GB/s (%.3f of 8 TB/s)" % (T, B, nch, ms, nb / ms / 1e6, nb / ms / 8e6 / 1e3), flush=True)
