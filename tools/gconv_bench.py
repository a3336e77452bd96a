"""Grouped 3x3 conv (+2x BN/ReLU) kernel vs MIOpen conv + 2 bn_relu passes, DPN92 stage shapes, B=32."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
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


for C, H in ((96, 128), (192, 64), (384, 32), (768, 16)):
    x = torch.randn(B, C, H, H, device=dev)
    w = torch.randn(C, C // 32, 3, 3, device=dev) * 0.1
    wt = ops.gconv_weight_layout(w, 32)
    bn1 = (torch.randn(C, device=dev), torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev))
    bn2 = (torch.randn(C, device=dev), torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev))
    t_new = timeit(lambda: ops.gconv3x3_bn_relu(x, wt, 32, bn1, bn2))
    t_old = timeit(lambda: ops.bn_relu(F.conv2d(ops.bn_relu(x, *bn1), w, None, 1, 1, 1, 32), *bn2))
    nb = 8 * B * C * H * H
    print("C=%d %dx%d cg=%d: fused %.3f ms (%.0f GB/s, %.1f%%)   miopen+2 passes %.3f ms"
          % (C, H, H, C // 32, t_new, nb / t_new / 1e6, nb / t_new / 1e6 / 80, t_old), flush=True)
