"""mix_bcast_add forward: store policy (plain / streamed) and pixels per workgroup, at the training shape
(B = 64, S = 100, M = 4020), the inference shapes that still use it (B < 6, S = 256) and the unfolded width."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dms_amd import ops
dev = "cuda"
L, K = 20, 10
def timeit(fn, iters=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters
for (B, S, M) in ((64, 100, 4020), (64, 100, 1000), (32, 256, 4020)):
    A = torch.randn(B, S, M, device=dev); Bt = torch.randn(L, B, M, device=dev)
    P = torch.randn(L, B, K, S, device=dev); W = torch.randn(M, K, device=dev)
    nb = 4 * B * (S * M + L * M + L * K * S + S * L * M)
    q = torch.empty(S * L, B, M, device=dev)
    ms = timeit(lambda: q.fill_(1.0)); print("B=%d S=%d M=%d fill_ of Q: %.4f ms %.0f GB/s" % (B, S, M, ms, q.numel() * 4 / ms / 1e6), flush=True)
    del q
    for pt in ("8", "4"):
        for nt in ("0", "1"):
            os.environ["DMN_MIX_FWD_PT"] = pt; os.environ["DMN_MIX_FWD_NT"] = nt
            ms = timeit(lambda: ops.mix_bcast_add(A, Bt, P, W))
            print("  PT=%s nt=%s: %.4f ms %.0f GB/s (%.1f%%)" % (pt, nt, ms, nb / ms / 1e6, nb / ms / 1e6 / 80), flush=True)
    del A, Bt, P, W
