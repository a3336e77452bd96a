#!/usr/bin/env python
"""dmn_dynfilter_fwd with COLD operands (a 1 GB fill between calls evicts L2 / Infinity Cache), as inside the step:
the repeat-the-same-call microbench keeps the 157 MB of operands cache-resident and flatters latency-bound variants."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dms_amd import ops
dev = "cuda"
L, K, C = 20, 10, 2696
big = torch.empty(256 * 1024 * 1024, device=dev)
for (B, S) in [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or ((32, 256), (64, 256), (64, 100), (16, 256)):
    g = torch.Generator(device=dev).manual_seed(B)
    V = torch.randn((B, C, S), device=dev, generator=g)
    F = torch.rand((L, B, K, C), device=dev, generator=g)
    fl = 2.0 * B * L * K * C * S
    out = []
    for bm, split, dma in (("64", "0", "0"), ("64", "1", "0"), ("32", "0", "0"), ("32", "1", "0"), ("64", "0", "4")):
        if True:
            os.environ["DMN_DYNFILTER_BM"] = bm; os.environ["DMN_DYNFILTER_SPLIT"] = split; os.environ["DMN_DYNFILTER_MFMA"] = "1"
            os.environ["DMN_DYNFILTER_DMA"] = "0" if dma == "0" else "1"; os.environ["DMN_DYNFILTER_DMA_D"] = dma
            for _ in range(2):
                ops.dynfilter(V, F)
            tot = 0.0
            for _ in range(10):
                big.fill_(1.0)
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); ops.dynfilter(V, F); b.record(); torch.cuda.synchronize()
                tot += a.elapsed_time(b)
            ms = tot / 10
            out.append("%s-row %s %.4f ms (%.1f TF)" % (bm, "DMA ring D=" + dma if dma != "0" else "split" if split == "1" else "whole", ms, fl / ms / 1e9))
    print("B=%d S=%d cold: " % (B, S) + " | ".join(out), flush=True)
