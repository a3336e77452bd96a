"""GPU parity tests (-m gpu), part 3: BASELINE configs[2]'s per-GPU shards and configs[3]'s glue
backward kernels at their REAL sizes.

  configs[2]  global batch 256 over 4 / 2 GPUs = 64 / 128 (image, query) pairs per GPU: the 512 px
              high-res forward at B = 64 and B = 128 on ONE GPU, samples inside each batch against the
              CPU oracle exactly as tests/test_gpu_parity_full.py does for B = 32 (the 8-GPU shard).
              B = 128 quadruples every buffer (layer-1 U: 7.9 GB, 92 % of 2^31 elements).
  configs[3]  dmn_mix_bcast_add_bwd (rows of 1000 and of 4020 floats) and dmn_dynfilter_bwd at B = 64,
              S = 100, L = 20, K = 10 (train.py:275-290) vs float64 restatements on the same inputs.
"""
import os

import pytest
import torch

import dms_amd
from dms_amd import _lib, ops
from oracle import dmn_ref, sru_ref
from tests.util import assert_close_elementwise, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), "these tests need a real MI355X"
    _lib.load()


@pytest.fixture(scope="module")
def dmn512():
    torch.manual_seed(1111)                                       # reference default seed (train.py:75)
    net = dms_amd.DMN(1000, num_filters=10, lang_layers=3, mix_we=True, pretrained=False,
                      high_res=True, upsampling_mode="bilinear", upsampling_amplification=32).eval()
    P = {k: v.clone() for k, v in net.state_dict().items()}
    cpu_base = net.langvis.base
    import copy
    cpu_base = copy.deepcopy(cpu_base)                            # stays on the CPU for the oracle
    return net.to(DEV), P, cpu_base


@pytest.mark.timeout(600)
@pytest.mark.parametrize("B,picks", [(64, (3, 63)), (128, (64, 127))])
def test_config3_shard_of_batch256_highres_512px_matches_cpu_oracle_inside_the_batch(dmn512, B, picks):
    """The 4-GPU (B = 64) and 2-GPU (B = 128) per-rank workloads of configs[2] on one GPU: two samples of
    the batch against torch-CPU DPN92 + oracle/dmn_ref + oracle upsampling.  Logits within 1e-4 relative
    (norm-wise AND element-wise), thresholded masks >= 99.9 % (north_star)."""
    net, P, cpu_base = dmn512
    g = torch.Generator().manual_seed(2000 + B)
    img = torch.randn((B, 3, 512, 512), generator=g)
    words = torch.randint(0, 1000, (B, 20), generator=g)
    cfg = dict(num_filters=10, vis_size=2688, lang_layers=3, mixed_layers=3, mix_we=True, high_res=True)
    sru_ref.set_threads(min(os.cpu_count() or 1, 16))
    want = {}
    with torch.no_grad():
        for b in picks:
            vis, feats = cpu_base(img[b:b + 1])
            x = dmn_ref.hot_path_single(P, vis, words[b:b + 1], cfg, use_c=True)
            want[b] = dmn_ref.upsampling(P, x, feats, 5)
    with torch.no_grad():
        out = net(img.to(DEV), words.to(DEV))
        finite = bool(torch.isfinite(out).all())
        got = {b: out[b:b + 1].cpu() for b in picks}
        same = bool(torch.equal(out[picks[0]], out[picks[1]]))
    assert tuple(out.shape) == (B, 1, 512, 512) and finite
    del out
    torch.cuda.empty_cache()
    for b in picks:
        assert rel_err(got[b], want[b]) < 1e-4, b
        assert_close_elementwise(got[b], want[b])
        agree = ((got[b] > 0) == (want[b] > 0)).float().mean().item()
        assert agree >= 0.999, (b, agree)
    assert not same


@pytest.mark.parametrize("M", [1000, 4020])
def test_config4_mix_bcast_add_bwd_B64_S100_full_size_matches_float64(M):
    """dmn_mix_bcast_add_bwd at the training step's own size (gQ = 2000 x 64 x M floats; M = 4020 is the
    folded layer-0 width): gA, gBt, gP on samples spread over the batch and gWp over the WHOLE batch vs
    float64 sums of the same gQ."""
    B, L, K, S = 64, 20, 10, 100
    gen = torch.Generator(device=DEV).manual_seed(4100 + M)
    P = torch.randn((L, B, K, S), device=DEV, generator=gen)
    Wp = 0.3 * torch.randn((M, K), device=DEV, generator=gen)
    gQ = torch.randn((S * L, B, M), device=DEV, generator=gen)
    A = torch.zeros((B, S, M), device=DEV).requires_grad_(True)
    Bt = torch.zeros((L, B, M), device=DEV).requires_grad_(True)
    Pg, Wg = P.clone().requires_grad_(True), Wp.clone().requires_grad_(True)
    ops.mix_bcast_add(A, Bt, Pg, Wg).backward(gQ)
    g4 = gQ.view(S, L, B, M)
    for b in (0, 1, 31, 62, 63):
        gb = g4[:, :, b].double()                                   # (S, L, M)
        assert rel_err(A.grad[b].cpu(), gb.sum(1).cpu()) < 1e-5, b
        assert rel_err(Bt.grad[:, b].cpu(), gb.sum(0).cpu()) < 1e-5, b
        gp = torch.einsum("mk,slm->lks", Wp.double(), gb)           # (L, K, S)
        assert rel_err(Pg.grad[:, b].cpu(), gp.cpu()) < 1e-5, b
    gw = torch.zeros((M, K), dtype=torch.float64, device=DEV)
    for b in range(B):
        gw += torch.einsum("lks,slm->mk", P[:, b].double(), g4[:, :, b].double())
    assert rel_err(Wg.grad.cpu(), gw.cpu()) < 1e-5


def test_config4_dynfilter_bwd_B64_S100_full_size_matches_float64():
    """dmn_dynfilter_bwd at the training step's own size (B = 64, 20 words x 10 filters x 2696 channels,
    S = 100): gV and gF for the whole batch vs the float64 einsum."""
    B, L, K, C, S = 64, 20, 10, 2696, 100
    gen = torch.Generator(device=DEV).manual_seed(4200)
    V = torch.randn((B, C, S), device=DEV, generator=gen)
    Fi = torch.rand((L, B, K, C), device=DEV, generator=gen)
    gP = torch.randn((L, B, K, S), device=DEV, generator=gen)
    Vg, Fg = V.clone().requires_grad_(True), Fi.clone().requires_grad_(True)
    ops.dynfilter(Vg, Fg).backward(gP)
    gV = torch.einsum("tbkc,tbks->bcs", Fi.double(), gP.double())
    gF = torch.einsum("tbks,bcs->tbkc", gP.double(), V.double())
    assert rel_err(Vg.grad.cpu(), gV.cpu()) < 1e-5
    assert rel_err(Fg.grad.cpu(), gF.cpu()) < 1e-5
    for b in (0, 63):                                               # and sample by sample (a norm over the batch hides one)
        assert rel_err(Vg.grad[b].cpu(), gV[b].cpu()) < 1e-5 and rel_err(Fg.grad[:, b].cpu(), gF[:, b].cpu()) < 1e-5
