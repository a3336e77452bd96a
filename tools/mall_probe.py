"""mSRU layer 1 at B=32: projection GEMM + scan over the whole sequence vs in time chunks that
alternate between two small U buffers (does the chunk stay in the 256 MB Infinity Cache?)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dms_amd import ops

dev = "cuda:0"
T, B, d, k = 5120, 32, 1005, 3
torch.manual_seed(0)
h0 = torch.randn(T, B, d, device=dev)
W = torch.randn(d, d * k, device=dev) * 0.03
bias = torch.randn(2 * d, device=dev) * 0.1
torch.backends.cuda.preferred_blas_library("cublas")


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def whole():
    u = torch.mm(h0.view(T * B, d), W).view(T, B, d, k)
    h, _, c = ops.sru_forward_raw(u, h0, bias, None, "tanh", 1.0, nchunks=1)
    return h, c


def gemm_only():
    return torch.mm(h0.view(T * B, d), W)


def chunked(tc):
    bufs = [torch.empty(tc * B, d * k, device=dev) for _ in range(2)]
    hs, c = [], None
    for i, t0 in enumerate(range(0, T, tc)):
        u = bufs[i % 2]
        torch.mm(h0[t0:t0 + tc].view(tc * B, d), W, out=u)
        h, _, c = ops.sru_forward_raw(u.view(tc, B, d, k), h0[t0:t0 + tc], bias, c, "tanh", 1.0, nchunks=1)
        hs.append(h)
    return hs, c


hw, cw = whole()
print("whole: gemm+scan %.3f ms (gemm alone %.3f ms)" % (timeit(whole), timeit(gemm_only)), flush=True)
for tc in (1024, 512, 256, 128):
    hs, c = chunked(tc)
    same = torch.equal(torch.cat(hs), hw) and torch.equal(c, cw)
    print("chunks of %4d steps (U chunk %.0f MB): %.3f ms  bit-identical=%s"
          % (tc, tc * B * d * k * 4 / 1e6, timeit(lambda: chunked(tc)), same), flush=True)
