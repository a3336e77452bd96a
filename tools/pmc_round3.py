#!/usr/bin/env python
"""A few launches of round 3's new / rewritten kernels at their real shapes, for rocprofv3 --pmc FETCH_SIZE /
WRITE_SIZE passes (separate runs) and --kernel-trace --stats:
  dmn_mix_bcast_add_bwd (one-pass kernel)   configs[3] training shape: B = 64, S = 100, L = 20, K = 10, M = 4020
  dmn_dynfilter_fwd (MFMA, ping-pong)       B = 32, L*K = 200, C = 2696, S = 256
  dmn_dynfilter_bwd                         B = 64, S = 100 (training shape)
  dmn_sru_fwd_image                         the 256-step tail of the last mSRU scan at B = 32, stored as the (B, Hm, H, W) map
  dmn_bias_sigmoid                          the filter Linear's bias + sigmoid, (B*L) x (K*C)
usage: python tools/pmc_round3.py [mixbwd|dyn|image|all]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dms_amd import _lib, ops
which = sys.argv[1] if len(sys.argv) > 1 else "all"
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
REPS = 3
st = torch.cuda.current_stream().cuda_stream
if which in ("mixbwd", "all"):
    B, L, K, S, M = 64, 20, 10, 100, 4020
    gQ = torch.randn((S * L, B, M), device=dev, generator=g)
    P = torch.randn((L, B, K, S), device=dev, generator=g)
    Wp = torch.randn((M, K), device=dev, generator=g)
    gA = torch.empty((B, S, M), device=dev); gBt = torch.empty((L, B, M), device=dev); gP = torch.empty_like(P)
    parts = _lib.query("dmn_mix_bwd_parts", B, L, K, S, M)
    nws = _lib.query("dmn_mix_bwd_workspace_floats", B, L, K, S, M)
    gW = torch.empty((parts, M, K), device=dev); ws = torch.empty((max(nws, 1),), device=dev)
    for _ in range(REPS):
        _lib.call("dmn_mix_bcast_add_bwd", gQ.data_ptr(), P.data_ptr(), Wp.data_ptr(), K, gA.data_ptr(), gBt.data_ptr(),
                  gP.data_ptr(), gW.data_ptr(), ws.data_ptr(), B, L, K, S, M, st)
    torch.cuda.synchronize()
    del gQ, P, gA, gBt, gP, gW, ws
if which in ("dyn", "all"):
    L, K, C = 20, 10, 2696
    V = torch.randn((32, C, 256), device=dev, generator=g); F = torch.rand((L, 32, K, C), device=dev, generator=g)
    for _ in range(REPS):
        ops.dynfilter(V, F)
    lin = torch.randn((32 * L, K * C), device=dev, generator=g); b = torch.randn((K * C,), device=dev, generator=g)
    for _ in range(REPS):
        ops.bias_sigmoid_(lin, b)
    torch.cuda.synchronize()
    B, S = 64, 100
    V = torch.randn((B, C, S), device=dev, generator=g); F = torch.rand((L, B, K, C), device=dev, generator=g)
    gP = torch.randn((L, B, K, S), device=dev, generator=g); gV = torch.empty_like(V); gF = torch.empty_like(F)
    for _ in range(REPS):
        _lib.call("dmn_dynfilter_bwd", V.data_ptr(), F.data_ptr(), gP.data_ptr(), gV.data_ptr(), gF.data_ptr(), B, L, K, C, S, st)
    torch.cuda.synchronize()
    del V, F, gP, gV, gF, lin
if which in ("image", "all"):
    T, B, d, Hs = 256, 32, 1005, 16
    u = torch.randn((T, B, d, 3), device=dev, generator=g); x = torch.randn((T, B, d), device=dev, generator=g)
    bias = 0.1 * torch.randn((2 * d,), device=dev, generator=g); c0 = torch.randn((B, d), device=dev, generator=g)
    whole = torch.empty((B, d + 8 + 1000, T), device=dev)
    for _ in range(REPS):
        ops.sru_forward_image(u, x, bias, c0, "tanh", 1.0, whole[:, :d])
    torch.cuda.synchronize()
