"""Race check for the two-wave fused mix + scan kernels: the same launch N times, every result compared bit for bit
with the first (and the first with mix + scan).  usage: python tools/mix_sru_soak.py [N]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dms_amd import ops

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = "cuda"
L, K = 20, 10
for (B, S, d) in ((32, 256, 1005), (64, 256, 1005), (16, 100, 1005), (7, 37, 130)):
    g = torch.Generator(device=dev).manual_seed(B)
    A = torch.randn((B, S, 4 * d), device=dev, generator=g)
    Bt = torch.randn((L, B, 4 * d), device=dev, generator=g)
    P = torch.randn((L, B, K, S), device=dev, generator=g)
    W = torch.randn((4 * d, K), device=dev, generator=g) * 0.1
    bias = torch.randn((2 * d,), device=dev, generator=g) * 0.1
    u = ops.mix_bcast_add(A, Bt, P, W).view(S * L, B, d, 4)
    h_ref, _, c_ref = ops.sru_forward_raw(u, None, bias, None, "tanh", 1.0, nchunks=1)
    del u
    for force in ("1", "0"):
        os.environ["DMN_MIX_SRU_MFMA"] = force
        bad = 0
        for i in range(N):
            h, c = ops.mix_sru_fwd(A, Bt, P, W, bias, None, "tanh")
            if not (torch.equal(h, h_ref) and torch.equal(c, c_ref)):
                bad += 1
        print("B=%d S=%d d=%d  DMN_MIX_SRU_MFMA=%s: %d launches, %d differ from mix + scan" % (B, S, d, force, N, bad), flush=True)
        assert bad == 0
    del os.environ["DMN_MIX_SRU_MFMA"]
print("ok")
