"""Few launches of the grouped conv kernel at the DPN92 conv2 / conv4 shapes for rocprofv3 --pmc passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dms_amd import ops
dev = "cuda"
for C, H in ((96, 128), (384, 32)):
    x = torch.randn(32, C, H, H, device=dev)
    w = torch.randn(C, C // 32, 3, 3, device=dev) * 0.1
    wt = ops.gconv_weight_layout(w, 32)
    bn1 = (torch.randn(C, device=dev), torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev))
    bn2 = (torch.randn(C, device=dev), torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev))
    for _ in range(3):
        ops.gconv3x3_bn_relu(x, wt, 32, bn1, bn2)
    torch.cuda.synchronize()
