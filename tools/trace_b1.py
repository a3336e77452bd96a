#!/usr/bin/env python
"""Batch-1 512 px high-res query, eager, a few dozen times (for rocprofv3 --kernel-trace)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from dms_amd import ops
dev = torch.device("cuda", 0)
if hasattr(ops, "use_gemm_table"):
    ops.use_gemm_table()
net = bench.build_model(dev, high_res=True)
g = torch.Generator(device=dev).manual_seed(3)
img = torch.randn((1, 3, 512, 512), device=dev, generator=g)
words = torch.randint(0, 1000, (1, 20), device=dev, generator=g)
with torch.no_grad():
    for _ in range(5):
        net(img, words)
    torch.cuda.synchronize()
    print("STEADY", flush=True)
    for _ in range(20):
        net(img, words)
    torch.cuda.synchronize()
