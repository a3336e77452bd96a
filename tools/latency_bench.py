"""Batch-B 512 px forward latency: eager launches vs one captured HIP graph replay."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dms_amd

dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
torch.manual_seed(1111)
net = dms_amd.DMN(1000, num_filters=10, lang_layers=3, mix_we=True, pretrained=False, high_res=True,
                  upsampling_mode="bilinear", upsampling_amplification=32).to(dev).eval()
img = torch.randn(B, 3, 512, 512, device=dev)
words = torch.randint(0, 1000, (B, 20), device=dev)


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


with torch.no_grad():
    ref = net(img, words)
    print("eager  B=%d: %.3f ms" % (B, timeit(lambda: net(img, words))), flush=True)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            net(img, words)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = net(img, words)
    g.replay()
    torch.cuda.synchronize()
    print("graph equal to eager:", bool(torch.equal(out, ref)), float((out - ref).abs().max()), flush=True)
    print("graph  B=%d: %.3f ms" % (B, timeit(lambda: g.replay())), flush=True)
