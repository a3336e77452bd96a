#!/usr/bin/env python
"""Probe (not used by the product): fp32 GEMM emulated by bf16 three-way operand splits -- a = a0 + a1 + a2,
the six largest cross products in ONE bf16 GEMM with K' = 6K and fp32 accumulation/output -- against the fp32
library GEMM, at the mSRU projection shape.  Reports time and error vs a float64 reference on a sub-block."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = "cuda"
M, K, N = 163840, 1005, 3015
g = torch.Generator(device=dev).manual_seed(1)
a = torch.randn((M, K), device=dev, generator=g)
b = torch.randn((K, N), device=dev, generator=g) * 0.03


def split3(x):
    x0 = x.to(torch.bfloat16); r = x - x0.float()
    x1 = r.to(torch.bfloat16); r = r - x1.float()
    return x0, x1, r.to(torch.bfloat16)


def timeit(fn, n=5):
    for _ in range(2):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


ms32 = timeit(lambda: torch.mm(a, b))
print("fp32 library GEMM %dx%dx%d: %.3f ms  %.0f TFLOP/s" % (M, K, N, ms32, 2e-9 * M * K * N / ms32), flush=True)
a0, a1, a2 = split3(a); b0, b1, b2 = split3(b)
A6 = torch.cat([a0, a0, a1, a0, a1, a2], 1).contiguous()            # (M, 6K)
B6 = torch.cat([b0, b1, b0, b2, b1, b0], 0).contiguous()            # (6K, N)
ms_split = timeit(lambda: torch.cat([t for t in split3(a)] * 2, 1))
try:
    c6 = torch.mm(A6, B6, out_dtype=torch.float32)
    ms6 = timeit(lambda: torch.mm(A6, B6, out_dtype=torch.float32))
    print("bf16 GEMM K' = 6K, fp32 out: %.3f ms  %.0f TFLOP/s (bf16)   [+ split/concat of A by torch ops: %.3f ms]" % (
        ms6, 2e-9 * M * 6 * K * N / ms6, ms_split), flush=True)
except Exception as ex:                                               # out_dtype unsupported on this build
    print("torch.mm(out_dtype=float32) on bf16 inputs failed:", repr(ex)[:300], flush=True)
    c6 = None
ms16 = timeit(lambda: torch.mm(a0, b0))
print("plain bf16 GEMM (bf16 out): %.3f ms  %.0f TFLOP/s" % (ms16, 2e-9 * M * K * N / ms16), flush=True)
rows = slice(0, 2048)
ref = a[rows].double() @ b.double()
c32 = torch.mm(a, b)[rows].double()
den = ref.abs().max()
print("max |err| / max |ref|: fp32 GEMM %.2e" % float((c32 - ref).abs().max() / den), flush=True)
if c6 is not None:
    print("max |err| / max |ref|: bf16x3 (6 products) %.2e" % float((c6[rows].double() - ref).abs().max() / den), flush=True)
