#!/usr/bin/env python
"""dmn_mix_bcast_add_bwd at the training shape (configs[3]: 320 px, B = 64): ms and achieved GB/s of the
two kernels together, register-weight pixel kernel (round 2) vs the LDS-weight one."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dms_amd import _lib
dev = "cuda"
B, L, K, S, M = (int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else (64, 20, 10, 100, 1000)))
g = torch.Generator(device=dev).manual_seed(1)
gQ = torch.randn((S * L, B, M), device=dev, generator=g)
P = torch.randn((L, B, K, S), device=dev, generator=g)
Wp = torch.randn((M, K), device=dev, generator=g)
gA = torch.empty((B, S, M), device=dev); gBt = torch.empty((L, B, M), device=dev)
gP = torch.empty_like(P); gW = torch.empty((L, B, M, K), device=dev)
st = torch.cuda.current_stream().cuda_stream
nbytes = 4 * (2 * gQ.numel() + gA.numel() + gBt.numel() + gP.numel() + gW.numel() + P.numel())


def run():
    _lib.call("dmn_mix_bcast_add_bwd", gQ.data_ptr(), P.data_ptr(), Wp.data_ptr(), K, gA.data_ptr(), gBt.data_ptr(),
              gP.data_ptr(), gW.data_ptr(), B, L, K, S, M, st)


for mode in ("00", "10", "11"):
    os.environ["DMN_MIX_BWD_PIX2"] = mode[0]
    os.environ["DMN_MIX_BWD_WORD2"] = mode[1]
    for _ in range(3):
        run()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        run()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    print("mix_bcast_add_bwd B=%d L=%d K=%d S=%d M=%d pix2/word2=%s: %.4f ms  %.0f GB/s (gQ read twice: %.0f MB)" % (
        B, L, K, S, M, mode, ms, nbytes / ms / 1e6, nbytes / 1e6), flush=True)
