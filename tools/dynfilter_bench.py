#!/usr/bin/env python
"""dmn_dynfilter_fwd at the 512 px shape (L*K = 200 filter rows, C = 2696, S = 256) over batch sizes:
one workgroup per tile vs one per (tile, chunk of C) + reduction."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dms_amd import ops
dev = "cuda"
L, K, C, S = 20, 10, 2696, 256
for B in (1, 2, 4, 8, 16, 32, 64):
    g = torch.Generator(device=dev).manual_seed(B)
    V = torch.randn((B, C, S), device=dev, generator=g)
    F = torch.rand((L, B, K, C), device=dev, generator=g)
    res = []
    for mode in ("0", "1"):
        os.environ["DMN_DYNFILTER_SPLIT"] = mode
        for _ in range(3):
            ops.dynfilter(V, F)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            ops.dynfilter(V, F)
        b.record(); torch.cuda.synchronize()
        res.append(a.elapsed_time(b) / 20)
    fl = 2.0 * B * L * K * C * S
    print("B=%2d: whole tiles %.4f ms (%.1f TFLOP/s)   split %.4f ms (%.1f TFLOP/s)" % (
        B, res[0], fl / res[0] / 1e9, res[1], fl / res[1] / 1e9), flush=True)
