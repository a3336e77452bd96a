#!/usr/bin/env python
"""dmn_sru_bwd at the configs[3] shapes (T = 2000 = 10x10 pixels x 20 words, B = 64, d = 1005):
HIP-event timing per ring depth (DMN_SRU_BWD_DEPTH), algorithmic 40 B per element vs 8 TB/s.
usage: python tools/sru_bwd_bench.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dms_amd import _lib, ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = "cuda"
T, B, d = 2000, 64, 1005
g = torch.Generator(device=dev).manual_seed(1)
for k in (3, 4):
    u = torch.randn((T, B, d, k), device=dev, generator=g)
    x = torch.randn((T, B, d), device=dev, generator=g) if k == 3 else None
    bias = 0.1 * torch.randn((2 * d,), device=dev, generator=g)
    c0 = torch.randn((B, d), device=dev, generator=g)
    gh = torch.randn((T, B, d), device=dev, generator=g)
    gl = torch.randn((B, d), device=dev, generator=g)
    _, c_all, _ = ops.sru_forward_raw(u, x, bias, c0, "tanh", 1.0, 0, want_c=True, nchunks=1)
    gu = torch.empty_like(u); gx = torch.empty_like(gh) if k == 3 else None
    gb = torch.empty(2 * d, device=dev); gc0 = torch.empty((B, d), device=dev)
    s = torch.cuda.current_stream().cuda_stream
    P = lambda t: None if t is None else t.data_ptr()
    nbytes = 40 * T * B * d
    for depth in ("2", "3", "4", "6", "8", "12", "16"):
        os.environ["DMN_SRU_BWD_DEPTH"] = depth
        call = lambda: _lib.call("dmn_sru_bwd", P(u), P(x), P(bias), P(c0), P(c_all), P(gh), P(gl), P(gu), P(gx),
                                 P(gb), P(gc0), None, T, B, d, k, 1, 1.0, s)
        for _ in range(3):
            call()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            call()
        b.record(); b.synchronize()
        ms = a.elapsed_time(b) / reps
        print("sru_bwd k=%d T=%d cols=%d depth=%s: %.4f ms  %.0f GB/s  %.1f %% of 8 TB/s"
              % (k, T, B * d, depth, ms, nbytes / ms / 1e6, nbytes / ms / 1e6 / 80), flush=True)
