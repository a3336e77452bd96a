"""High-res head alone at B=32: UpsamplingModule with tap_gemm, then winograd = False / "f2" / "auto" / True."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dms_amd

dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
torch.manual_seed(0)
up = dms_amd.UpsamplingModule(1005, mode="bilinear", ker_size=3, amplification=32).to(dev).eval()
x = torch.randn(B, 1005, 16, 16, device=dev)
feats = [torch.randn(B, c, s, s, device=dev) for c, s in ((64, 256), (336, 128), (704, 64), (1552, 32), (2688, 16))]


def timeit(fn, n=5):
    for _ in range(2):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


with torch.no_grad():
    up.tap_gemm = False
    up.winograd = False
    ref = up(x, feats)
    up.tap_gemm = True
    y = up(x, feats)
    err = ((y - ref).abs().max() / ref.abs().max()).item()
    print("tap_gemm        %.2f ms   rel diff vs MIOpen path %.2e" % (timeit(lambda: up(x, feats)), err), flush=True)
    up.tap_gemm = False
    for mode in (False, "f2", "auto", True):
        up.winograd = mode
        y = up(x, feats)
        err = ((y - ref).abs().max() / ref.abs().max()).item()
        print("winograd=%-5s  %.2f ms   rel diff vs MIOpen path %.2e" % (mode, timeit(lambda: up(x, feats)), err), flush=True)
