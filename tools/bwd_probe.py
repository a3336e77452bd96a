#!/usr/bin/env python
"""Backward glue kernels at the config-4 shapes (B=64, 320px) for rocprofv3 --kernel-trace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dms_amd import ops
dev = "cuda"
B, L, K, S, M, C8 = 64, 20, 10, 100, 1000, 2696
A = torch.randn(B, S, M, device=dev, requires_grad=True); Bt = torch.randn(L, B, M, device=dev, requires_grad=True)
P = torch.randn(L, B, K, S, device=dev, requires_grad=True); W = torch.randn(M, 4706, device=dev, requires_grad=True)
gQ = torch.randn(S * L, B, M, device=dev)
V = torch.randn(B, C8, S, device=dev, requires_grad=True); Fi = torch.rand(L, B, K, C8, device=dev, requires_grad=True)
gP = torch.randn(L, B, K, S, device=dev)
for _ in range(5):
    ops.mix_bcast_add(A, Bt, P, W[:, 4696:]).backward(gQ)
    ops.dynfilter(V, Fi).backward(gP)
torch.cuda.synchronize()
