"""tap_upsample_gather / tap_project alone at the B=32 head shapes: ms and achieved GB/s."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dms_amd import ops

dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


tot = 0.0
for co, s in ((512, 16), (256, 32), (128, 64), (64, 128), (1, 256)):
    Z = torch.randn(B, 9, co, s, s, device=dev)
    bias = torch.randn(co, device=dev)
    ms = timeit(lambda: ops.tap_upsample_gather(Z, bias))
    nbytes = 4 * B * co * s * s * 13
    tot += ms
    print("gather Cout=%4d %3dx%-3d  %.4f ms  %.0f GB/s" % (co, s, s, ms, nbytes / ms / 1e6), flush=True)
print("gather total %.3f ms" % tot)
x = torch.randn(B, 64, 256, 256, device=dev)
f = torch.randn(B, 64, 256, 256, device=dev)
w9 = torch.randn(9, 128, device=dev)
ms = timeit(lambda: ops.tap_project(x, f, w9))
print("project 128 -> 9 rows 256x256  %.4f ms  %.0f GB/s" % (ms, 4 * B * 65536 * 137 / ms / 1e6))
