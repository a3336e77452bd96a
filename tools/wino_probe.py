"""Probe: non-fused Winograd F(2x2,3x3) for the head's 3x3 convs: cost of the 16 batched fp32
GEMMs (rocBLAS) vs MIOpen's conv, and accuracy of a torch prototype of the transforms."""
import sys
import time

import torch
import torch.nn.functional as F

dev = "cuda:0"
B = 32
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


BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32, device=dev)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float32, device=dev)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32, device=dev)

levels = [(3693, 512, 32), (2064, 256, 64), (960, 128, 128), (464, 64, 256)]
for cin, cout, H in levels:
    W = H
    nt = B * (H // 2) * (W // 2)
    w = torch.randn(cout, cin, 3, 3, device=dev) * 0.02
    U = torch.einsum("ij,ocjk,lk->iloc", G, w, G).reshape(16, cout, cin).contiguous()
    V = torch.randn(16, cin, nt, device=dev)
    t_bmm = timeit(lambda: torch.bmm(U, V))
    fl = 2.0 * 16 * cout * cin * nt
    print("cin %4d cout %3d %3dx%-3d  bmm(16) %.2f ms (%.0f TF actual, %.0f TF conv-equivalent)  V %.2f GB"
          % (cin, cout, H, W, t_bmm, fl / t_bmm / 1e9, fl * 2.25 / t_bmm / 1e9, V.numel() * 4 / 1e9), flush=True)
    del V
    if H <= 64:     # accuracy of the algorithm (torch prototype, small batch)
        x = torch.randn(2, cin, H, W, device=dev)
        ref = F.conv2d(x, w, None, 1, 1)
        xp = F.pad(x, (1, 1, 1, 1))
        tiles = xp.unfold(2, 4, 2).unfold(3, 4, 2)                  # (2, C, H/2, W/2, 4, 4)
        Vt = torch.einsum("ij,bcyxjk,lk->ilcbyx", BT, tiles, BT).reshape(16, cin, -1)
        M = torch.bmm(U, Vt).view(4, 4, cout, 2, H // 2, W // 2)
        Y = torch.einsum("ij,jkobyx,lk->boyixl", AT, M, AT).reshape(2, cout, H, W)
        print("   winograd vs direct: max rel err %.2e" % ((Y - ref).abs().max() / ref.abs().max()).item(), flush=True)
        del x, xp, tiles, Vt, M, Y
    torch.cuda.empty_cache()
