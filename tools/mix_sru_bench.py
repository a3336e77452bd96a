"""Layer 0 of the mSRU with the folded comb_conv: mix kernel + scan kernel vs the fused kernel."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dms_amd import ops

dev = "cuda"
L, K, S, d = 20, 10, 256, 1005


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


for B in (2, 4, 8, 12, 16, 32, 64):
    A = torch.randn(B, S, 4 * d, device=dev)
    Bt = torch.randn(L, B, 4 * d, device=dev)
    P = torch.randn(L, B, K, S, device=dev)
    W = torch.randn(4 * d, K, device=dev) * 0.1
    bias = torch.randn(2 * d, device=dev) * 0.1

    def pair():
        u = ops.mix_bcast_add(A, Bt, P, W).view(S * L, B, d, 4)
        return ops.sru_forward_raw(u, None, bias, None, "tanh", 1.0)

    t_pair = timeit(pair)
    # the fused call includes the (L,B,K,S) -> (B,S,L,K) transpose of P (a small torch copy)
    os.environ["DMN_MIX_SRU_PAIR"] = "0"
    t_single = timeit(lambda: ops.mix_sru_fwd(A, Bt, P, W, bias, None, "tanh"))
    os.environ["DMN_MIX_SRU_PAIR"] = "1"
    os.environ["DMN_MIX_SRU_MFMA"] = "0"
    t_two = timeit(lambda: ops.mix_sru_fwd(A, Bt, P, W, bias, None, "tanh"))
    h_two = ops.mix_sru_fwd(A, Bt, P, W, bias, None, "tanh")[0]
    os.environ["DMN_MIX_SRU_MFMA"] = "1"
    t_mfma = timeit(lambda: ops.mix_sru_fwd(A, Bt, P, W, bias, None, "tanh"))
    h_mfma = ops.mix_sru_fwd(A, Bt, P, W, bias, None, "tanh")[0]
    del os.environ["DMN_MIX_SRU_MFMA"]
    t_auto = timeit(lambda: ops.mix_sru_fwd(A, Bt, P, W, bias, None, "tanh"))
    same = torch.equal(h_two, pair()[0])
    print("B=%2d: mix + scan %.3f ms   fused single-wave %.3f   two-wave VALU %.3f   two-wave MFMA %.3f   by itself %.3f ms   "
          "VALU == mix + scan %s   MFMA == VALU %s"
          % (B, t_pair, t_single, t_two, t_mfma, t_auto, same, torch.equal(h_mfma, h_two)), flush=True)
    del A, Bt, P, W
