#!/usr/bin/env python
"""Few launches of the in-situ mSRU scan shapes for rocprofv3 --pmc passes (HBM traffic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dms_amd import ops
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
for (T, B, d, k, h_t0) in [(5120, 32, 1005, 4, 0), (5120, 32, 1005, 3, 0), (5120, 32, 1005, 3, 4864), (4096, 128, 1000, 3, 0)]:
    u = torch.randn((T, B, d, k), device=dev, generator=g)
    x = torch.randn((T, B, d), device=dev, generator=g) if k == 3 else None
    bias = 0.1 * torch.randn((2 * d,), device=dev, generator=g)
    for _ in range(3):
        ops.sru_forward_raw(u, x, bias, None, "tanh", 1.0, h_t0, want_c=False, nchunks=1)
    torch.cuda.synchronize()
    del u, x
# state-only head of the last mSRU layer (u0 | forget columns)
u2 = torch.randn((4864, 32, 1005, 2), device=dev, generator=g)
bias = 0.1 * torch.randn((2 * 1005,), device=dev, generator=g)
for _ in range(3):
    ops.sru_state(u2, bias)
torch.cuda.synchronize()
