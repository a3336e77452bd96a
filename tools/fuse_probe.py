"""Probe: MIOpen fused conv+bias+relu (aten::miopen_convolution_relu) vs conv + dmn_bn_relu
for the DPN92 shapes at B=32 / 512 px.  Prints ms per call."""
import sys
import time

import torch

sys.path.insert(0, ".")
from dms_amd import ops  # noqa: E402

dev = "cuda:0"
B = 32


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


shapes = [  # name, cin, cout, k, stride, groups, H
    ("c2 1x1a", 336, 96, 1, 1, 1, 128), ("c2 3x3b", 96, 96, 3, 1, 32, 128), ("c2 1x1c", 96, 272, 1, 1, 1, 128),
    ("c3 1x1a", 704, 192, 1, 1, 1, 64), ("c3 3x3b", 192, 192, 3, 1, 32, 64), ("c3 1x1c", 192, 544, 1, 1, 1, 64),
    ("c4 1x1a", 1552, 384, 1, 1, 1, 32), ("c4 3x3b", 384, 384, 3, 1, 32, 32), ("c4 1x1c", 384, 1048, 1, 1, 1, 32),
    ("c5 1x1a", 2560, 768, 1, 1, 1, 16), ("c5 3x3b", 768, 768, 3, 1, 32, 16), ("c5 1x1c", 768, 2176, 1, 1, 1, 16),
]
for name, cin, cout, k, stride, groups, H in shapes:
    x = torch.randn(B, cin, H, H, device=dev)
    w = torch.randn(cout, cin // groups, k, k, device=dev) * 0.05
    bias = torch.randn(cout, device=dev)
    mean = torch.randn(cout, device=dev)
    scale = torch.rand(cout, device=dev) + 0.5
    pad = k // 2
    with torch.no_grad():
        t_conv = timeit(lambda: torch.nn.functional.conv2d(x, w, None, stride, pad, 1, groups))
        t_sep = timeit(lambda: ops.bn_relu(torch.nn.functional.conv2d(x, w, None, stride, pad, 1, groups),
                                           mean, scale, bias))
        try:
            t_fused = timeit(lambda: torch.ops.aten.miopen_convolution_relu(
                x, w, bias, [stride, stride], [pad, pad], [1, 1], groups))
            y1 = torch.ops.aten.miopen_convolution_relu(x, w, bias, [stride, stride], [pad, pad], [1, 1], groups)
            y2 = torch.relu(torch.nn.functional.conv2d(x, w, bias, stride, pad, 1, groups))
            err = ((y1 - y2).abs().max() / y2.abs().max()).item()
        except Exception as e:  # noqa: BLE001
            t_fused, err = float("nan"), str(e)[:80]
    print("%-8s conv %.3f  conv+bn_relu %.3f  miopen_fused %.3f  err %s" % (name, t_conv, t_sep, t_fused, err),
          flush=True)
