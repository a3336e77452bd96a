"""mix_bcast_add at the folded layer-0 shape (M = 4020) and the plain comb_conv shape (M = 1000)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dms_amd import ops

dev = "cuda"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


L, K = 20, 10
S = int(sys.argv[2]) if len(sys.argv) > 2 else 256
for M, mode in ((4020, "0"), (4020, "1"), (1000, "0"), (1000, "1")):
    os.environ["DMN_MIX_FWD_PK"] = mode
    A = torch.randn(B, S, M, device=dev)
    Bt = torch.randn(L, B, M, device=dev)
    P = torch.randn(L, B, K, S, device=dev)
    W = torch.randn(M, K, device=dev)
    nb = 4 * B * (S * M + L * M + L * K * S + S * L * M)
    ms = timeit(lambda: ops.mix_bcast_add(A, Bt, P, W))
    print("B=%d S=%d pk=%s M=%d: %.3f ms %.0f GB/s (%.1f%%)" % (B, S, mode, M, ms, nb / ms / 1e6, nb / ms / 1e6 / 80), flush=True)
    del A, Bt, P, W
