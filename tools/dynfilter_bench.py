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
    for mode in ("0", "1", "0m", "1m", "0n", "1n"):          # m: matrix cores, 64-row tiles; n: 32-row tiles
        os.environ["DMN_DYNFILTER_SPLIT"] = mode[0]
        os.environ["DMN_DYNFILTER_MFMA"] = "1" if mode[-1] in "mn" else "0"
        os.environ["DMN_DYNFILTER_BM"] = "32" if mode.endswith("n") else "64"
        for _ in range(3):
            ops.dynfilter(V, F)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            ops.dynfilter(V, F)
        b.record(); torch.cuda.synchronize()
        res.append(a.elapsed_time(b) / 20)
    fl = 2.0 * B * L * K * C * S
    print("B=%2d: packed FMA whole %.4f ms (%.1f TF) split %.4f ms (%.1f TF) | MFMA 64-row whole %.4f ms (%.1f TF) split %.4f ms (%.1f TF) | MFMA 32-row whole %.4f ms (%.1f TF) split %.4f ms (%.1f TF)" % (
        B, res[0], fl / res[0] / 1e9, res[1], fl / res[1] / 1e9, res[2], fl / res[2] / 1e9, res[3], fl / res[3] / 1e9,
        res[4], fl / res[4] / 1e9, res[5], fl / res[5] / 1e9), flush=True)

# forward at the training shape (S = 100, B = 64)
Bt, St = 64, 100
g = torch.Generator(device=dev).manual_seed(5)
V = torch.randn((Bt, C, St), device=dev, generator=g); F = torch.rand((L, Bt, K, C), device=dev, generator=g)
for bm in ("64", "32"):
    os.environ["DMN_DYNFILTER_MFMA"] = "1"; os.environ["DMN_DYNFILTER_BM"] = bm; os.environ.pop("DMN_DYNFILTER_SPLIT", None)
    for _ in range(3):
        ops.dynfilter(V, F)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        ops.dynfilter(V, F)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    print("dynfilter_fwd B=%d S=%d MFMA %s-row tiles: %.4f ms  %.1f TFLOP/s" % (Bt, St, bm, ms, 2.0 * Bt * L * K * C * St / ms / 1e9), flush=True)
del V, F
os.environ.pop("DMN_DYNFILTER_BM", None)

# backward at the training shape (configs[3]: 320 px low-res, S = 100, B = 64)
from dms_amd import _lib
B, S = 64, 100
g = torch.Generator(device=dev).manual_seed(7)
V = torch.randn((B, C, S), device=dev, generator=g); F = torch.rand((L, B, K, C), device=dev, generator=g)
gP = torch.randn((L, B, K, S), device=dev, generator=g)
gV = torch.empty_like(V); gF = torch.empty_like(F)
st = torch.cuda.current_stream().cuda_stream
for mode in ("0", "1"):
    os.environ["DMN_DYNFILTER_BWD_VEC"] = mode
    fn = lambda: _lib.call("dmn_dynfilter_bwd", V.data_ptr(), F.data_ptr(), gP.data_ptr(), gV.data_ptr(), gF.data_ptr(), B, L, K, C, S, st)
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        fn()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    print("dynfilter_bwd B=%d S=%d vec=%s: %.4f ms  %.1f TFLOP/s" % (B, S, mode, ms, 4.0 * B * L * K * C * S / ms / 1e9), flush=True)
