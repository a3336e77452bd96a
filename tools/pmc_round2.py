#!/usr/bin/env python
"""A few launches of round 2's new / changed HBM-side kernels at their real shapes, for
rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs) and --kernel-trace --stats:
  dmn_mix_sru_fwd (two-wave kernel)  B = 32, 512 px   algorithmic: h 646 MB written, A + Bt + P 160 MB read
  dmn_sru_bwd k = 3 / k = 4          T = 2000, B = 64, d = 1005 (configs[3])   40 B per element
  dmn_tap_upsample_gather            the five head levels at B = 32
usage: python tools/pmc_round2.py [mix|bwd|tap|all]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dms_amd import _lib, ops
which = sys.argv[1] if len(sys.argv) > 1 else "all"
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
REPS = 3
if which in ("mix", "all"):
    B, L, K, S, d = 32, 20, 10, 256, 1005
    A = torch.randn((B, S, 4 * d), device=dev, generator=g)
    Bt = torch.randn((L, B, 4 * d), device=dev, generator=g)
    P = torch.randn((L, B, K, S), device=dev, generator=g)
    W = torch.randn((4 * d, K), device=dev, generator=g) * 0.1
    bias = 0.1 * torch.randn((2 * d,), device=dev, generator=g)
    for _ in range(REPS):
        ops.mix_sru_fwd(A, Bt, P, W, bias, None, "tanh")
    torch.cuda.synchronize()
    del A, Bt, P, W
if which in ("bwd", "all"):
    T, B, d = 2000, 64, 1005
    for k in (3, 4):
        u = torch.randn((T, B, d, k), device=dev, generator=g)
        x = torch.randn((T, B, d), device=dev, generator=g) if k == 3 else None
        bias = 0.1 * torch.randn((2 * d,), device=dev, generator=g)
        gh = torch.randn((T, B, d), device=dev, generator=g)
        _, c_all, _ = ops.sru_forward_raw(u, x, bias, None, "tanh", 1.0, 0, want_c=True, nchunks=1)
        gu = torch.empty_like(u); gx = torch.empty_like(gh) if k == 3 else None
        gb = torch.empty(2 * d, device=dev); gc0 = torch.empty((B, d), device=dev)
        s = torch.cuda.current_stream().cuda_stream
        Pp = lambda t: None if t is None else t.data_ptr()
        for _ in range(REPS):
            _lib.call("dmn_sru_bwd", Pp(u), Pp(x), Pp(bias), None, Pp(c_all), Pp(gh), None, Pp(gu), Pp(gx),
                      Pp(gb), Pp(gc0), None, T, B, d, k, 1, 1.0, s)
        torch.cuda.synchronize()
        del u, x, gh, c_all, gu, gx
if which in ("tap", "all"):
    B = 32
    for co, hw in ((512, 16), (256, 32), (128, 64), (64, 128), (1, 256)):
        Z = torch.randn((B, 9, co, hw, hw), device=dev, generator=g)
        bias = torch.randn((co,), device=dev, generator=g)
        for _ in range(REPS):
            ops.tap_upsample_gather(Z, bias)
        torch.cuda.synchronize()
        del Z
