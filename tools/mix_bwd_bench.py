#!/usr/bin/env python
"""dmn_mix_bcast_add_bwd at the training shape (configs[3]: 320 px, B = 64): ms and achieved GB/s of the
one-pass kernels (round 3: 256- and 512-column workgroups, with and without the XCD-aware block order) vs the
two-pass kernels (round 2), at the unfolded (M = 1000) and folded (M = 4020) row widths.
GB/s counts gQ ONCE (the algorithmic traffic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dms_amd import _lib
dev = "cuda"
B, L, K, S, M = (int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else (64, 20, 10, 100, 4020)))
g = torch.Generator(device=dev).manual_seed(1)
gQ = torch.randn((S * L, B, M), device=dev, generator=g)
P = torch.randn((L, B, K, S), device=dev, generator=g)
Wp = torch.randn((M, K), device=dev, generator=g)
gA = torch.empty((B, S, M), device=dev); gBt = torch.empty((L, B, M), device=dev)
gP = torch.empty_like(P)
st = torch.cuda.current_stream().cuda_stream
nbytes = 4 * (gQ.numel() + gA.numel() + gBt.numel() + gP.numel() + B * M * K + P.numel())
REPS = int(os.environ.get("MIXBENCH_REPS", "20"))
NAMES = {"0": "two passes", "1": "ONE pass, 256-column workgroups", "2": "ONE pass, 512-column workgroups"}


def bench(mode, xcd):
    os.environ["DMN_MIX_BWD_ONE"] = "0" if mode == "0" else "1"
    os.environ["DMN_MIX_ONE_HALVES"] = mode if mode in "12" else "1"
    os.environ["DMN_MIX_ONE_XCD"] = xcd
    parts = _lib.query("dmn_mix_bwd_parts", B, L, K, S, M)
    nws = _lib.query("dmn_mix_bwd_workspace_floats", B, L, K, S, M)
    gW = torch.empty((parts, M, K), device=dev)
    ws = torch.empty((max(nws, 1),), device=dev)

    def run():
        _lib.call("dmn_mix_bcast_add_bwd", gQ.data_ptr(), P.data_ptr(), Wp.data_ptr(), K, gA.data_ptr(), gBt.data_ptr(),
                  gP.data_ptr(), gW.data_ptr(), ws.data_ptr(), B, L, K, S, M, st)
        return gW.sum(0)

    for _ in range(3):
        run()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(REPS):
        run()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / REPS
    print("mix_bcast_add_bwd (+ the caller's sum over %d gWp slabs) B=%d L=%d K=%d S=%d M=%d %s%s: %.4f ms  %.0f GB/s = %.3f of 8 TB/s "
          "(algorithmic %.0f MB, gQ once)" % (parts, B, L, K, S, M, NAMES[mode], "" if mode == "0" else " xcd_map=" + xcd, ms,
                                             nbytes / ms / 1e6, nbytes / ms / 1e6 / 8000, nbytes / 1e6), flush=True)


for mode in os.environ.get("MIXBENCH_MODES", "012"):
    for xcd in (("1",) if mode == "0" else ("0", "1")):
        bench(mode, xcd)
