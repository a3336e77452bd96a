#!/usr/bin/env python
"""A few launches of this round's new HBM-bound kernels at their B=32 / 512 px shapes, for
rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (HBM traffic vs algorithmic bytes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dms_amd import ops
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
B = 32
# DPN conv4 stage, a middle block: Co = 1024 + 48 + 11*24
c1c, inc, m = 1024, 24, 11
Co = c1c + 2 * inc + m * inc
buf = torch.randn((B, Co + inc + 9 * inc, 32, 32), device=dev, generator=g)
bn = tuple(torch.randn((Co,), device=dev, generator=g) for _ in range(3))
for _ in range(3):
    ops.dpn_stage_pass(buf, c1c, inc, Co, bn)
torch.cuda.synchronize()
del buf
Z = torch.randn((B, 9, 64, 128, 128), device=dev, generator=g)
bias = torch.randn((64,), device=dev, generator=g)
for _ in range(3):
    ops.tap_upsample_gather(Z, bias)
torch.cuda.synchronize()
del Z
x = torch.randn((B, 64, 256, 256), device=dev, generator=g)
f = torch.randn((B, 64, 256, 256), device=dev, generator=g)
w9 = torch.randn((9, 128), device=dev, generator=g)
for _ in range(3):
    ops.tap_project(x, f, w9)
mean, scale, b2 = (torch.randn((64,), device=dev, generator=g) for _ in range(3))
for _ in range(3):
    ops.bn_relu_maxpool(x, mean, scale, b2)
torch.cuda.synchronize()
del x, f
L, K, S, d = 20, 10, 256, 1005
A = torch.randn((B, S, 4 * d), device=dev, generator=g)
Bt = torch.randn((L, B, 4 * d), device=dev, generator=g)
P = torch.randn((L, B, K, S), device=dev, generator=g)
W = torch.randn((4 * d, K), device=dev, generator=g) * 0.1
bias = 0.1 * torch.randn((2 * d,), device=dev, generator=g)
for _ in range(3):
    ops.mix_sru_fwd(A, Bt, P, W, bias, None, "tanh")
torch.cuda.synchronize()
