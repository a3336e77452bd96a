#!/usr/bin/env python
"""Debug aid: dmn_sru_bwd (and the training forward) are deterministic -- repeat launches at the
configs[3] size must agree bit for bit in EVERY element; report where they do not."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dms_amd import _lib, ops
DEV = "cuda"
k = int(sys.argv[1]) if len(sys.argv) > 1 else 3
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
T, B, d = 2000, 64, 1005
gen = torch.Generator(device=DEV).manual_seed(2222 + k)
u = torch.randn((T, B, d, k), device=DEV, generator=gen)
x = torch.randn((T, B, d), device=DEV, generator=gen) if k == 3 else None
bias = 0.1 * torch.randn((2 * d,), device=DEV, generator=gen)
c0 = torch.randn((B, d), device=DEV, generator=gen)
gh = torch.randn((T, B, d), device=DEV, generator=gen)
gl = torch.randn((B, d), device=DEV, generator=gen)
P = lambda t: None if t is None else t.data_ptr()
s = torch.cuda.current_stream().cuda_stream


def fwd():
    return ops.sru_forward_raw(u, x, bias, c0, "tanh", 1.0, 0, want_c=True, nchunks=1)


def bwd(c_all):
    gu = torch.empty_like(u); gx = torch.empty_like(gh) if k == 3 else None
    gb = torch.empty(2 * d, device=DEV); gc0 = torch.empty((B, d), device=DEV)
    _lib.call("dmn_sru_bwd", P(u), P(x), P(bias), P(c0), P(c_all), P(gh), P(gl), P(gu), P(gx), P(gb), P(gc0), None,
              T, B, d, k, 1, 1.0, s)
    return gu, gx, gc0


def where(a, b, name):
    ne = (a != b)
    if a.dim() == 4:
        ne = ne.any(-1)
    ne = ne.view(T, B * d)
    tt, cc = torch.nonzero(ne, as_tuple=True)
    waves = sorted(set((cc // 64).tolist()))
    print("   %s differs: %d elements, t %d..%d, %d columns, %d waves %s..%s, lanes %s" % (
        name, int(ne.sum()), int(tt.min()), int(tt.max()), len(set(cc.tolist())), len(waves), waves[:6], waves[-6:],
        sorted(set((cc % 64).tolist()))[:70]), flush=True)
    if a.dim() == 4:
        comp = (a != b).view(T, B * d, -1)[tt, cc].sum(0).tolist()
        i = 0
        av, bv = a.view(T, B * d, -1), b.view(T, B * d, -1)
        print("      per component:", comp, "| sample t=%d col=%d this=%s first=%s | first at t+1: %s t-1: %s" % (
            int(tt[i]), int(cc[i]), av[tt[i], cc[i]].tolist(), bv[tt[i], cc[i]].tolist(),
            bv[min(int(tt[i]) + 1, T - 1), cc[i]].tolist(), bv[max(int(tt[i]) - 1, 0), cc[i]].tolist()), flush=True)


h0, c0_all, _ = fwd()
g0 = bwd(c0_all)
nbad_f = nbad_b = 0
for r in range(reps):
    h1, c1_all, _ = fwd()
    if not (torch.equal(h1, h0) and torch.equal(c1_all, c0_all)):
        nbad_f += 1
        where(h1, h0, "forward h") if not torch.equal(h1, h0) else where(c1_all, c0_all, "forward c")
    g1 = bwd(c0_all)
    if not torch.equal(g1[0], g0[0]):
        nbad_b += 1
        where(g1[0], g0[0], "backward grad_U")
print("k=%d: %d repeats, forward differed %d times, backward %d times" % (k, reps, nbad_f, nbad_b), flush=True)
