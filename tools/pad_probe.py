"""Would running the mSRU stack at d = 1008 (zero-padded) instead of 1005 pay?  The projection GEMM
with K, N multiples of 16, and the scan at the padded width, both with TunableOp's best solution."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.cuda.tunable as tunable
from dms_amd import ops

dev = "cuda:0"
T, B = 5120, 32
tunable.enable(True)
tunable.tuning_enable(True)
tunable.set_max_tuning_duration(30)
tunable.set_max_tuning_iterations(20)
tunable.set_filename("/tmp/pad_probe_tunable.csv")


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


for d in (1005, 1008):
    for k in (3, 2):
        rows = T * B if k == 3 else (T - 256) * B
        h = torch.randn(rows, d, device=dev)
        W = torch.randn(d, d * k, device=dev) * 0.03
        ms = timeit(lambda: torch.mm(h, W))
        print("GEMM (%d x %d) . (%d x %d): %.3f ms  %.1f TFLOP/s (useful-flop basis d=1005: %.1f)"
              % (rows, d, d, d * k, ms, 2 * rows * d * d * k / ms / 1e9, 2 * rows * 1005 * 1005 * k / ms / 1e9), flush=True)
        del h, W
    for k in (4, 3):
        u = torch.randn(T, B, d, k, device=dev)
        x = torch.randn(T, B, d, device=dev) if k == 3 else None
        bias = torch.randn(2 * d, device=dev) * 0.1
        ms = timeit(lambda: ops.sru_forward_raw(u, x, bias, None, "tanh", 1.0, nchunks=1))
        print("scan d=%d k=%d: %.3f ms" % (d, k, ms), flush=True)
        del u, x
L, K, S = 20, 10, 256
for M in (4020, 4032):
    A = torch.randn(B, S, M, device=dev)
    Bt = torch.randn(L, B, M, device=dev)
    P = torch.randn(L, B, K, S, device=dev)
    Wp = torch.randn(M, K, device=dev)
    print("mix M=%d: %.3f ms" % (M, timeit(lambda: ops.mix_bcast_add(A, Bt, P, Wp))), flush=True)
    del A, Bt, P, Wp
