"""Probe: the five 3x3 convolutions of the high-res head at B=32 (dmn.py:315-326 shapes):
MIOpen (NCHW, what the model runs) vs nine shifted GEMMs on a zero-padded NHWC tensor."""
import sys
import time

import torch
import torch.nn.functional as F

dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
torch.backends.cuda.preferred_blas_library("cublas")


def timeit(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


levels = [(3693, 512, 32), (2064, 256, 64), (960, 128, 128), (464, 64, 256), (128, 32, 512)]
for cin, cout, H in levels:
    W = H
    x = torch.randn(B, cin, H, W, device=dev)
    w = torch.randn(cout, cin, 3, 3, device=dev) * 0.02
    bias = torch.randn(cout, device=dev)
    flops = 2.0 * B * H * W * cin * cout * 9
    with torch.no_grad():
        t_mi = timeit(lambda: F.conv2d(x, w, bias, 1, 1))
        ref = F.conv2d(x, w, bias, 1, 1)
        # padded NHWC, flat pixel rows
        xp = F.pad(x.permute(0, 2, 3, 1), (0, 0, 1, 1, 1, 1)).contiguous()     # (B, H+2, W+2, C)
        flat = xp.view(-1, cin)
        Wp = W + 2
        M = flat.shape[0] - (2 * Wp + 2)
        wk = [w[:, :, dy, dx].t().contiguous() for dy in range(3) for dx in range(3)]   # (cin, cout)
        offs = [dy * Wp + dx for dy in range(3) for dx in range(3)]
        out = torch.empty(M, cout, device=dev)

        def nine():
            torch.addmm(bias, flat[offs[0]:offs[0] + M], wk[0], out=out)
            for k in range(1, 9):
                out.addmm_(flat[offs[k]:offs[k] + M], wk[k])
            return out

        t_9 = timeit(nine)
        got = torch.zeros(B * (H + 2) * Wp, cout, device=dev)
        got[:M] = nine()
        got = got.view(B, H + 2, Wp, cout)[:, :H, :W].permute(0, 3, 1, 2)
        err = ((got - ref).abs().max() / ref.abs().max()).item()
        # one GEMM with K = 9*cin on a gathered (im2col-free) view is not expressible; also try
        # channels_last MIOpen for reference
        xcl = x.contiguous(memory_format=torch.channels_last)
        wcl = w.contiguous(memory_format=torch.channels_last)
        t_cl = timeit(lambda: F.conv2d(xcl, wcl, bias, 1, 1))
    print("cin %4d cout %3d %3dx%-3d  miopen %.2f ms (%.0f TF)  nhwc-miopen %.2f ms  nine-gemm %.2f ms (%.0f TF) err %.1e"
          % (cin, cout, H, W, t_mi, flops / t_mi / 1e9, t_cl, t_9, flops / t_9 / 1e9, err), flush=True)
    del x, w, xp, flat, out, got, ref, xcl, wcl
    torch.cuda.empty_cache()
