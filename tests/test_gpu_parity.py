"""GPU parity tests (-m gpu): every HIP entry point, called through the C ABI, against the
CPU oracle on the same seeded inputs; the model-level path against the reference's golden
vectors; and size-independent properties at BASELINE.json's full sizes.

Tolerances: fp32, logits within 1e-4 relative (BASELINE.json north_star); kernels use
v_exp_f32 / v_rcp_f32 (1 ulp) so element-wise checks use rtol 1e-4 + atol 1e-5 at most
and in practice sit near 1e-6.
"""
import os
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import dms_amd
from dms_amd import _lib, ops
from oracle import dmn_ref, sru_ref
from oracle.detgen import det_fill_state, det_ints, det_uniform
from tests.util import assert_close_elementwise, base_inputs, cfg_from_golden, load_golden, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), "these tests need a real MI355X"
    _lib.load()


def _case(T, B, d, k, seed):
    u = det_uniform((T, B, d, k), seed, 1.5)
    x = det_uniform((T, B, d), seed + 1, 1.0) if k == 3 else None
    bias = det_uniform((2 * d,), seed + 2, 0.5)
    c0 = det_uniform((B, d), seed + 3, 1.0)
    return u, x, bias, c0


def _g(t):
    return None if t is None else t.to(DEV)


def _np(t):
    return None if t is None else t.numpy()


SRU_SHAPES = [(1, 1, 5, 3), (7, 3, 5, 4), (64, 1, 64, 3), (33, 2, 21, 4), (100, 3, 1005, 4),
              (100, 3, 1005, 3), (257, 1, 130, 3), (0, 2, 4, 3), (25, 1, 1, 4), (23, 2, 63, 3)]


ACTS = ["tanh", "relu", "identity"]     # upstream's use_tanh / use_relu switches (unknown default: all three run)
SCALES = [1.0, 1.7]                     # upstream's highway scale_x (applies to k == 3 only)


@pytest.mark.parametrize("k", [3, 4])
@pytest.mark.parametrize("want_c", [False, True])
def test_sru_fwd_wide_scan_in_column_ranges_equals_one_launch(k, want_c):
    """From 1536 wavefronts on a scan that stores every step is launched as consecutive column ranges
    (dmn_sru_fwd); the ranges must tile the columns exactly: same bits as one launch, ragged last range."""
    T, B, d = 9, 99, 1003                                     # 99 297 columns = 1552 wavefronts, the last one partial
    u = det_uniform((T, B, d, k), 901, 1.5).to(DEV)
    x = det_uniform((T, B, d), 902, 1.0).to(DEV) if k == 3 else None
    bias = det_uniform((2 * d,), 903, 0.5).to(DEV)
    c0 = det_uniform((B, d), 904, 1.0).to(DEV)
    outs = []
    for split in ("0", None, "3", "7"):
        if split is not None:
            os.environ["DMN_SRU_SPLIT"] = split
        try:
            outs.append(ops.sru_forward_raw(u, x, bias, c0, "tanh", 1.0, 0, want_c=want_c, nchunks=1))
        finally:
            os.environ.pop("DMN_SRU_SPLIT", None)
    h0, c_all0, cl0 = outs[0]
    for h, c_all, cl in outs[1:]:
        assert torch.equal(h, h0) and torch.equal(cl, cl0)
        if want_c:
            assert torch.equal(c_all, c_all0)
    # a few columns against the oracle (first, around the range boundaries of the default split, last)
    cols = [0, 1, 49663, 49664, 49665, B * d - 2, B * d - 1]
    uu = u.view(T, B * d, k)[:, cols].cpu().numpy().reshape(T, 1, len(cols), k)
    xx = None if x is None else x.view(T, B * d)[:, cols].cpu().numpy().reshape(T, 1, len(cols))
    jj = [c % d for c in cols]
    bb = np.concatenate([bias.cpu().numpy()[jj], bias.cpu().numpy()[[d + j for j in jj]]])
    h_o, _, cl_o = sru_ref.sru_fwd_c(uu, xx, bb, c0.view(-1)[cols].cpu().numpy().reshape(1, -1), "tanh", 1.0)
    np.testing.assert_allclose(h0.view(T, B * d)[:, cols].cpu().numpy(), h_o[:, 0], rtol=1e-4, atol=3e-6)
    np.testing.assert_allclose(cl0.view(-1)[cols].cpu().numpy(), cl_o[0], rtol=1e-4, atol=3e-6)


@pytest.mark.parametrize("T,B,d,k", SRU_SHAPES)
@pytest.mark.parametrize("act", ACTS)
@pytest.mark.parametrize("scale_x", SCALES)
def test_sru_fwd_matches_oracle(T, B, d, k, act, scale_x):
    u, x, bias, c0 = _case(T, B, d, k, 41)
    h_ref, c_ref, cl_ref = sru_ref.sru_fwd_c(_np(u), _np(x), _np(bias), _np(c0), act, scale_x)
    h, c_all, c_last = ops.sru_forward_raw(_g(u), _g(x), _g(bias), _g(c0), act, scale_x, 0,
                                           want_c=True, nchunks=1)
    np.testing.assert_allclose(h.cpu().numpy(), h_ref, rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(c_all.cpu().numpy(), c_ref, rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(c_last.cpu().numpy(), cl_ref, rtol=1e-4, atol=2e-6)
    # inference variant: no c store, c0 = NULL, only the tail of h
    h_t0 = T // 3
    h_ref2, _, cl2 = sru_ref.sru_fwd_c(_np(u), _np(x), _np(bias), None, act, scale_x, want_c=False)
    h2, none_c, c_last2 = ops.sru_forward_raw(_g(u), _g(x), _g(bias), None, act, scale_x, h_t0,
                                              want_c=False, nchunks=1)
    assert none_c is None and h2.shape == (T - h_t0, B, d)
    np.testing.assert_allclose(h2.cpu().numpy(), h_ref2[h_t0:], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(c_last2.cpu().numpy(), cl2, rtol=1e-4, atol=2e-6)


@pytest.mark.parametrize("depth", ["3", "4", "6", "8", "16", "24", "32"])
def test_sru_fwd_every_prefetch_depth(depth, monkeypatch):
    # DMN_SRU_DEPTH forces the register-ring depth; T % depth != 0 exercises the tails
    monkeypatch.setenv("DMN_SRU_DEPTH", depth)
    T, B, d, k = 61, 2, 70, 4
    u, x, bias, c0 = _case(T, B, d, k, 47)
    h_ref, _, _ = sru_ref.sru_fwd_c(_np(u), None, _np(bias), _np(c0), "tanh", 1.0)
    h, _, _ = ops.sru_forward_raw(_g(u), None, _g(bias), _g(c0), "tanh", 1.0, 0, nchunks=1)
    np.testing.assert_allclose(h.cpu().numpy(), h_ref, rtol=1e-4, atol=2e-6)
    u, x, bias, c0 = _case(T, B, d, 3, 48)
    h_ref, _, _ = sru_ref.sru_fwd_c(_np(u), _np(x), _np(bias), _np(c0), "tanh", 1.0)
    h, _, _ = ops.sru_forward_raw(_g(u), _g(x), _g(bias), _g(c0), "tanh", 1.0, 0, nchunks=1)
    np.testing.assert_allclose(h.cpu().numpy(), h_ref, rtol=1e-4, atol=2e-6)


@pytest.mark.parametrize("T,B,d,k,nch", [(640, 1, 1005, 4, 10), (640, 1, 1005, 3, 7),
                                         (513, 2, 33, 3, 8), (5, 1, 9, 4, 4), (70, 1, 64, 3, 70)])
@pytest.mark.parametrize("act,scale_x", [("tanh", 1.0), ("relu", 1.7), ("identity", 1.7), ("identity", 1.0)])
def test_sru_fwd_time_parallel_scan_matches_oracle(T, B, d, k, nch, act, scale_x):
    u, x, bias, c0 = _case(T, B, d, k, 53)
    h_ref, _, cl_ref = sru_ref.sru_fwd_c(_np(u), _np(x), _np(bias), _np(c0), act, scale_x)
    h_t0 = T // 2
    h, _, c_last = ops.sru_forward_raw(_g(u), _g(x), _g(bias), _g(c0), act, scale_x, h_t0,
                                       want_c=False, nchunks=nch)
    np.testing.assert_allclose(h.cpu().numpy(), h_ref[h_t0:], rtol=1e-4, atol=3e-6)
    np.testing.assert_allclose(c_last.cpu().numpy(), cl_ref, rtol=1e-4, atol=3e-6)


@pytest.mark.parametrize("T,B,d,k", [(1, 1, 5, 3), (9, 3, 5, 4), (40, 2, 17, 3), (100, 2, 1005, 4),
                                     (67, 3, 130, 3), (8, 1, 64, 4), (0, 1, 3, 3)])
@pytest.mark.parametrize("act", ACTS)
@pytest.mark.parametrize("scale_x", SCALES)
def test_sru_bwd_matches_oracle(T, B, d, k, act, scale_x):
    u, x, bias, c0 = _case(T, B, d, k, 61)
    gh = det_uniform((T, B, d), 67, 1.0)
    gl = det_uniform((B, d), 68, 1.0)
    _, c_all, _ = sru_ref.sru_fwd_c(_np(u), _np(x), _np(bias), _np(c0), act, scale_x)
    gu_r, gx_r, gb_r, gc0_r = sru_ref.sru_bwd_c(_np(u), _np(x), _np(bias), _np(c0), c_all,
                                                _np(gh), _np(gl), act, scale_x)
    ug, xg, bg, cg = (_g(u).requires_grad_(True), _g(x).requires_grad_(True) if x is not None else None,
                      _g(bias).requires_grad_(True), _g(c0).requires_grad_(True))
    h, c_last = ops.sru_compute(ug, xg, bg, cg, act, scale_x)
    ((h * _g(gh)).sum() + (c_last * _g(gl)).sum()).backward()
    np.testing.assert_allclose(ug.grad.cpu().numpy(), gu_r, rtol=2e-4, atol=1e-5)
    np.testing.assert_allclose(bg.grad.cpu().numpy(), gb_r.sum(1).reshape(-1), rtol=2e-4, atol=5e-5)
    np.testing.assert_allclose(cg.grad.cpu().numpy(), gc0_r, rtol=2e-4, atol=1e-5)
    if x is not None:
        np.testing.assert_allclose(xg.grad.cpu().numpy(), gx_r, rtol=2e-4, atol=1e-5)


@pytest.mark.parametrize("depth", ["2", "3", "4", "6", "8", "12", "16"])
@pytest.mark.parametrize("T", [1, 2, 5, 16, 37, 48])
def test_sru_bwd_every_ring_depth_and_sequence_length(depth, T, monkeypatch):
    """DMN_SRU_BWD_DEPTH forces the backward ring depth: T below, equal to and above a multiple of it
    (prologue-only, exactly one block, clamped last block, partial tail), k = 3 and k = 4."""
    monkeypatch.setenv("DMN_SRU_BWD_DEPTH", depth)
    for k in (3, 4):
        B, d = 2, 70
        u, x, bias, c0 = _case(T, B, d, k, 63)
        gh, gl = det_uniform((T, B, d), 64, 1.0), det_uniform((B, d), 65, 1.0)
        _, c_all, _ = sru_ref.sru_fwd_c(_np(u), _np(x), _np(bias), _np(c0), "tanh", 1.0)
        gu_r, gx_r, gb_r, gc0_r = sru_ref.sru_bwd_c(_np(u), _np(x), _np(bias), _np(c0), c_all, _np(gh), _np(gl), "tanh", 1.0)
        ug, bg, cg = (_g(t).requires_grad_(True) for t in (u, bias, c0))
        xg = _g(x).requires_grad_(True) if x is not None else None
        h, c_last = ops.sru_compute(ug, xg, bg, cg, "tanh", 1.0)
        ((h * _g(gh)).sum() + (c_last * _g(gl)).sum()).backward()
        np.testing.assert_allclose(ug.grad.cpu().numpy(), gu_r, rtol=2e-4, atol=1e-5)
        np.testing.assert_allclose(bg.grad.cpu().numpy(), gb_r.sum(1).reshape(-1), rtol=2e-4, atol=5e-5)
        np.testing.assert_allclose(cg.grad.cpu().numpy(), gc0_r, rtol=2e-4, atol=1e-5)
        if x is not None:
            np.testing.assert_allclose(xg.grad.cpu().numpy(), gx_r, rtol=2e-4, atol=1e-5)


@pytest.mark.parametrize("depth", ["3", "4", "6", "8", "16"])
@pytest.mark.parametrize("T", [1, 2, 5, 16, 37, 48])
def test_sru_state_scan_backward_matches_the_full_scan_backward(depth, T, monkeypatch):
    """Training form of the state-only scan (dmn_sru_state_fwd_train / dmn_sru_state_bwd, k = 2 ring):
    c_last bit-identical to the full k = 3 scan's, and the gradients w.r.t. [u0 | forget], the forget
    bias and c0 equal those of the full scan when nothing reads its hidden states (grad_h = 0), at every
    ring depth and for T below / at / above a multiple of it."""
    monkeypatch.setenv("DMN_SRU_BWD_DEPTH", depth)
    B, d, k = 2, 70, 3
    u, x, bias, c0 = _case(T, B, d, k, 163)
    gl = det_uniform((B, d), 165, 1.0)
    ug, bg, cg = (_g(t).requires_grad_(True) for t in (u, bias, c0))
    h, c_full = ops.sru_compute(ug, _g(x), bg, cg, "tanh", 1.0)
    (c_full * _g(gl)).sum().backward()
    u2 = _g(u)[..., :2].contiguous().requires_grad_(True)
    b2, c2 = _g(bias).requires_grad_(True), _g(c0).requires_grad_(True)
    c_state = ops.sru_state_train(u2, b2, c2)
    assert torch.equal(c_state, c_full)
    (c_state * _g(gl)).sum().backward()
    np.testing.assert_allclose(u2.grad.cpu().numpy(), ug.grad[..., :2].cpu().numpy(), rtol=1e-5, atol=1e-6)
    assert float(ug.grad[..., 2].abs().max()) == 0.0              # the reset column carries no gradient here
    np.testing.assert_allclose(b2.grad[:d].cpu().numpy(), bg.grad[:d].cpu().numpy(), rtol=1e-4, atol=1e-5)
    assert float(b2.grad[d:].abs().max()) == 0.0
    np.testing.assert_allclose(c2.grad.cpu().numpy(), cg.grad.cpu().numpy(), rtol=1e-5, atol=1e-6)
    # without c0
    u3 = _g(u)[..., :2].contiguous().requires_grad_(True)
    c_n = ops.sru_state_train(u3, _g(bias), None)
    (c_n * _g(gl)).sum().backward()
    assert torch.equal(c_n, ops.sru_forward_raw(_g(u), _g(x), _g(bias), None, "tanh", 1.0, nchunks=1)[2])
    assert bool(torch.isfinite(u3.grad).all())


@pytest.mark.parametrize("n_in,n_out,T,tail", [(10, 10, 37, 5), (12, 10, 37, 5), (10, 10, 6, 5), (16, 16, 64, 1)])
def test_sru_module_tail_training_shortcut_equals_the_plain_autograd_path(n_in, n_out, T, tail):
    """SRU(..., tail=n) under autograd: the last layer's first T - n steps run on the [u0 | forget] columns
    with the state-only scan (SRUCell.state_only_head_in_training); output and EVERY gradient equal the
    plain path's (one full scan per layer) to rounding."""
    B = 3
    x = det_uniform((T, B, n_in), 171, 1.0)
    g = det_uniform((tail, B, n_out), 172, 1.0)
    got = {}
    for flag in (True, False):
        rnn = dms_amd.SRU(n_in, n_out, num_layers=3)
        rnn.load_state_dict(det_fill_state(rnn.state_dict()))
        rnn = rnn.to(DEV)
        for cell in rnn.rnn_lst:
            cell.state_only_head_in_training = flag
        xg = x.to(DEV).requires_grad_(True)
        out, c = rnn(xg, tail=tail)
        assert out.shape == (tail, B, n_out)
        ((out * g.to(DEV)).sum() + c.sum()).backward()
        got[flag] = [out.detach().cpu(), xg.grad.cpu()] + [p.grad.cpu() for p in rnn.parameters()]
    for a, b in zip(got[True], got[False]):
        assert rel_err(a, b) < 1e-5


@pytest.mark.parametrize("T,B,d", [(1, 1, 5), (7, 3, 5), (33, 2, 70), (300, 2, 1005), (64, 1, 64)])
@pytest.mark.parametrize("k", [3, 4])
def test_sru_state_only_scan_is_bit_identical_to_full_scan_and_matches_oracle(T, B, d, k):
    """dmn_sru_state_fwd on the [u0 | forget] columns == c_last of dmn_sru_fwd on all k."""
    u = det_uniform((T, B, d, k), 51, 2.0)
    x = det_uniform((T, B, d), 52, 1.0) if k == 3 else None
    bias = det_uniform((2 * d,), 53, 0.5)
    c0 = det_uniform((B, d), 54, 1.0)
    _, _, c_ref = sru_ref.sru_fwd_c(u.numpy(), None if x is None else x.numpy(), bias.numpy(), c0.numpy(), "tanh", 1.0)
    ud, bd, cd = u.to(DEV), bias.to(DEV), c0.to(DEV)
    _, _, c_full = ops.sru_forward_raw(ud, None if x is None else x.to(DEV), bd, cd, "tanh", 1.0, nchunks=1)
    u2 = ud[..., :2].contiguous()
    c_state = ops.sru_state(u2, bd, cd, nchunks=1)
    assert torch.equal(c_state, c_full)
    for nch in (2, 5, 64):                                  # time-parallel variant (re-associated)
        c_par = ops.sru_state(u2, bd, cd, nchunks=nch)
        np.testing.assert_allclose(c_par.cpu().numpy(), c_ref, rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(c_state.cpu().numpy(), c_ref, rtol=1e-5, atol=2e-6)
    assert torch.equal(ops.sru_state(u2, bd, None, nchunks=1),
                       ops.sru_forward_raw(ud, None if x is None else x.to(DEV), bd, None, "tanh", 1.0, nchunks=1)[2])


@pytest.mark.parametrize("depth", ["3", "4", "6", "8", "16", "24", "32"])
def test_sru_state_only_scan_every_prefetch_depth(depth, monkeypatch):
    monkeypatch.setenv("DMN_SRU_DEPTH", depth)
    T, B, d = 131, 2, 100
    u = det_uniform((T, B, d, 3), 55, 2.0).to(DEV)
    x = det_uniform((T, B, d), 56, 1.0).to(DEV)
    bias = det_uniform((2 * d,), 57, 0.5).to(DEV)
    c_full = ops.sru_forward_raw(u, x, bias, None, "tanh", 1.0, nchunks=1)[2]
    assert torch.equal(ops.sru_state(u[..., :2].contiguous(), bias, nchunks=1), c_full)


@pytest.mark.parametrize("n_in,n_out", [(24, 24), (16, 24)])
def test_sru_tail_shortcut_equals_full_scan(n_in, n_out):
    """Inference with tail=S: the last layer's head steps run state-only on 2 of k columns;
    result must equal the plain full-length scan (and so the oracle, tested above)."""
    rnn = dms_amd.SRU(n_in, n_out, num_layers=2)
    rnn.load_state_dict(det_fill_state(rnn.state_dict()))
    rnn = rnn.to(DEV).eval()
    x = det_uniform((90, 3, n_in), 58, 1.0).to(DEV)
    with torch.no_grad():
        h_fast, c_fast = rnn(x, tail=12)
        for cell in rnn.rnn_lst:
            cell.state_only_head = False
        h_full, c_full = rnn(x, tail=12)
        h_all, _ = rnn(x)
    assert h_fast.shape == (12, 3, n_out)
    assert rel_err(h_fast, h_full) < 1e-5 and rel_err(c_fast, c_full) < 1e-5
    assert rel_err(h_fast, h_all[-12:]) < 1e-5
    # with autograd the shortcut is not taken and gradients flow
    for cell in rnn.rnn_lst:
        cell.state_only_head = True
    h_g, _ = rnn(x, tail=12)
    assert h_g.requires_grad and rel_err(h_g.detach(), h_full) < 1e-5


def test_sru_module_stack_matches_oracle_and_autograd():
    torch.manual_seed(5)
    rnn = dms_amd.SRU(12, 20, num_layers=3).to(DEV)
    ref = sru_ref.SRURef(12, 20, num_layers=3)
    ref.load_state_dict({k: v.cpu() for k, v in rnn.state_dict().items()})
    x = det_uniform((17, 2, 12), 71, 1.0)
    xg = x.to(DEV).requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    h, c = rnn(xg)
    hr, cr = ref(xr)
    assert h.shape == (17, 2, 20) and c.shape == (3, 2, 20)
    assert rel_err(h.detach().cpu(), hr.detach()) < 1e-5 and rel_err(c.detach().cpu(), cr.detach()) < 1e-5
    w = det_uniform((17, 2, 20), 72, 1.0)
    (h * w.to(DEV)).sum().backward()
    (hr * w).sum().backward()
    assert rel_err(xg.grad.cpu(), xr.grad) < 1e-4
    for (n, p), (_, pr) in zip(rnn.named_parameters(), ref.named_parameters()):
        assert rel_err(p.grad.cpu(), pr.grad) < 1e-4, n
    with torch.no_grad():
        h_tail, _ = rnn(xg, tail=5)
    assert h_tail.shape == (5, 2, 20) and rel_err(h_tail.cpu(), hr.detach()[-5:]) < 1e-5


@pytest.mark.parametrize("H,W", [(10, 10), (16, 16), (16, 12), (1, 1), (3, 5)])
def test_cat_coords_bit_exact_vs_reference_golden(H, W):
    g = load_golden("spatial")["s_%d_%d" % (H, W)]
    vis = det_uniform((2, 5, H, W), 81, 1.0)
    V = ops.cat_coords(vis.to(DEV)).cpu()
    assert torch.equal(V[:, :5], vis)
    for b in range(2):
        np.testing.assert_array_equal(V[b, 5:].numpy(), g[0])
    s = dms_amd.BaseDMN.generate_spatial_batch(H, W).cpu().numpy()
    np.testing.assert_array_equal(s, g)


@pytest.mark.parametrize("B,L,K,C,S", [(1, 20, 10, 2696, 256), (2, 20, 10, 2696, 100), (3, 5, 3, 40, 12),
                                       (2, 4, 1, 37, 15), (1, 1, 1, 1, 1), (2, 7, 10, 130, 66)])
def test_dynfilter_matches_per_word_conv2d(B, L, K, C, S):
    V = det_uniform((B, C, S), 91, 1.0)
    Fi = torch.sigmoid(det_uniform((L, B, K, C), 92, 2.0))
    P = ops.dynfilter(V.to(DEV), Fi.to(DEV)).cpu()
    # the reference's formulation: one F.conv2d per word, per sample (dmn.py:178-184)
    ref = torch.stack([torch.cat([torch.nn.functional.conv2d(
        V[b].double().view(1, C, S, 1), Fi[t, b].double().view(K, C, 1, 1)).view(1, K, S)
        for b in range(B)]) for t in range(L)])
    assert P.shape == (L, B, K, S)
    assert rel_err(P, ref) < 1e-5       # 2696-long fp32 sums vs an fp64 reference


@pytest.mark.parametrize("B,L,K,C,S", [(1, 20, 10, 2696, 256), (2, 20, 10, 2696, 100), (3, 5, 3, 40, 12),
                                       (1, 2, 2, 16, 4), (2, 7, 9, 1000, 64), (5, 20, 10, 264, 256), (16, 20, 10, 72, 256),
                                       (2, 13, 10, 2696, 68)])
def test_dynfilter_split_contraction_is_bit_identical_to_the_whole_tile_kernel(B, L, K, C, S, monkeypatch):
    """One workgroup per (tile, chunk of C) + reduction (few samples) must give the SAME bits as one
    workgroup per tile (the chunk order of the summation is fixed), so that a sample's filter responses
    do not depend on the batch it arrives in; both against the float64 einsum."""
    V = det_uniform((B, C, S), 96, 1.0).to(DEV)
    Fi = torch.sigmoid(det_uniform((L, B, K, C), 97, 2.0)).to(DEV)
    ref = torch.einsum("tbkc,bcs->tbks", Fi.double().cpu(), V.double().cpu())
    res = {}
    monkeypatch.setenv("DMN_DYNFILTER_DMA", "0")               # (the LDS-DMA ring kernel has its own pass below)
    for mfma in ("1", "0"):                                   # matrix-core kernel / packed-FMA kernel
        monkeypatch.setenv("DMN_DYNFILTER_MFMA", mfma)
        monkeypatch.setenv("DMN_DYNFILTER_SPLIT", "1")
        split = ops.dynfilter(V, Fi)
        monkeypatch.setenv("DMN_DYNFILTER_SPLIT", "0")
        whole = ops.dynfilter(V, Fi)
        monkeypatch.delenv("DMN_DYNFILTER_SPLIT")
        auto = ops.dynfilter(V, Fi)
        assert torch.equal(split, whole) and torch.equal(auto, whole), mfma
        assert rel_err(whole.cpu(), ref) < 1e-5, mfma
        res[mfma] = whole
    monkeypatch.delenv("DMN_DYNFILTER_MFMA")
    # fp32 MFMA continues the same k-ordered fmaf chain: the two kernels agree bit for bit
    assert torch.equal(res["1"], res["0"])
    whole = res["1"]
    # ... and so does the kernel that streams both operands through an LDS-DMA ring (ragged rows / columns / C included:
    # rows past R and columns past S are fetched out of bounds, a last K-step past C multiplies zero feature rows)
    monkeypatch.setenv("DMN_DYNFILTER_DMA", "1")
    for bm in ("64", "32"):
        monkeypatch.setenv("DMN_DYNFILTER_BM", bm)
        guard = torch.full((L + 2, B, K, S), 7.0, device=DEV)
        rc = _lib.call("dmn_dynfilter_fwd", V.data_ptr(), Fi.data_ptr(), guard[1:-1].data_ptr(), B, L, K, C, S,
                       torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert torch.equal(guard[1:-1], whole), bm
        assert bool((guard[0] == 7.0).all()) and bool((guard[-1] == 7.0).all())
    monkeypatch.delenv("DMN_DYNFILTER_BM")
    monkeypatch.delenv("DMN_DYNFILTER_DMA")
    assert torch.equal(ops.dynfilter(V, Fi), whole)            # whatever the library picks by itself
    # a sample alone == the same sample inside the batch, bit for bit
    if B > 1:
        alone = ops.dynfilter(V[1:2].contiguous(), Fi[:, 1:2].contiguous())
        assert torch.equal(alone[:, 0], whole[:, 1])


@pytest.mark.parametrize("B,L,K,C,S", [(2, 5, 3, 41, 18), (2, 20, 10, 2696, 100), (1, 20, 10, 264, 256),
                                       (3, 7, 9, 40, 12), (1, 2, 2, 16, 4), (2, 3, 5, 68, 64)])
def test_dynfilter_bwd_matches_autograd(B, L, K, C, S, monkeypatch):
    """gV, gF vs float64 autograd of the einsum; the 16-byte-load kernels (C, S multiples of 4) and the
    element-wise guarded ones agree."""
    V = det_uniform((B, C, S), 93, 1.0)
    Fi = det_uniform((L, B, K, C), 94, 1.0)
    gP = det_uniform((L, B, K, S), 95, 1.0)
    Vr, Fr = V.clone().double().requires_grad_(True), Fi.clone().double().requires_grad_(True)
    (torch.einsum("tbkc,bcs->tbks", Fr, Vr) * gP.double()).sum().backward()
    got = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("DMN_DYNFILTER_BWD_VEC", mode)
        Vg, Fg = V.to(DEV).requires_grad_(True), Fi.to(DEV).requires_grad_(True)
        (ops.dynfilter(Vg, Fg) * gP.to(DEV)).sum().backward()
        assert rel_err(Vg.grad.cpu(), Vr.grad) < 1e-5 and rel_err(Fg.grad.cpu(), Fr.grad) < 1e-5, mode
        got[mode] = (Vg.grad.cpu(), Fg.grad.cpu())
    assert rel_err(got["1"][0], got["0"][0]) < 2e-6 and rel_err(got["1"][1], got["0"][1]) < 2e-6


def _mix_ref(A, Bt, P, Wp):
    # Q[(s*L+t), b, m]
    L, B, K, S = P.shape
    q = A.permute(1, 0, 2)[:, None] + Bt[None] + torch.einsum("mk,tbks->stbm", Wp, P)
    return q.reshape(S * L, B, -1)


@pytest.mark.parametrize("B,L,K,S,M", [(2, 20, 10, 100, 1000), (1, 5, 3, 12, 16), (3, 4, 1, 7, 10),
                                       (2, 3, 17, 9, 8), (1, 2, 20, 5, 7)])
def test_mix_bcast_add_matches_einsum(B, L, K, S, M):
    A = det_uniform((B, S, M), 101, 1.0)
    Bt = det_uniform((L, B, M), 102, 1.0)
    P = det_uniform((L, B, K, S), 103, 1.0)
    Wfull = det_uniform((M, 50 + K), 104, 1.0)
    Wp = Wfull.to(DEV)[:, 50:]                       # strided column slice like comb_conv's
    Q = ops.mix_bcast_add(A.to(DEV), Bt.to(DEV), P.to(DEV), Wp).cpu()
    ref = _mix_ref(A.double(), Bt.double(), P.double(), Wfull[:, 50:].double())
    assert Q.shape == (S * L, B, M)
    assert rel_err(Q, ref) < 1e-5


@pytest.mark.parametrize("B,L,K,S,M", [(4, 20, 10, 9, 4020),      # B*M % 16 == 0: shifted (64-byte aligned) windows, shifts 0/4/8/12
                                       (16, 20, 10, 3, 1000),     # aligned, one piece
                                       (3, 20, 10, 10, 4020),     # B*M % 16 != 0: unshifted
                                       (4, 5, 3, 17, 1028),       # K < 10, ragged pixel tile, last piece of 4 + shift
                                       (32, 2, 10, 2, 36)])       # rows narrower than a wave
def test_mix_bcast_add_packed_aligned_kernel_is_bit_identical_to_the_plain_one(B, L, K, S, M, monkeypatch):
    """mix_fwd_pk_kernel (v_pk_fma_f32, column windows on 64-byte lines) does the same fmaf chain per element
    as mix_fwd_kernel: torch.equal, and nothing outside Q is written (guard rows around the buffer)."""
    A, Bt = det_uniform((B, S, M), 131, 1.0).to(DEV), det_uniform((L, B, M), 132, 1.0).to(DEV)
    P, Wp = det_uniform((L, B, K, S), 133, 1.0).to(DEV), det_uniform((M, K), 134, 1.0).to(DEV)
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("DMN_MIX_FWD_PK", mode)
        out[mode] = ops.mix_bcast_add(A, Bt, P, Wp)
    assert torch.equal(out["0"], out["1"])
    for pt in ("4", "8"):                      # pixels per workgroup x store policy (streamed / plain)
        for nt in ("0", "1"):
            monkeypatch.setenv("DMN_MIX_FWD_PT", pt)
            monkeypatch.setenv("DMN_MIX_FWD_NT", nt)
            guard = torch.full((S * L + 2, B, M), 7.0, device=DEV)       # one guard row on either side of Q
            q = guard[1:-1]
            _lib.call("dmn_mix_bcast_add", A.data_ptr(), Bt.data_ptr(), P.data_ptr(), Wp.data_ptr(), Wp.stride(0),
                      q.data_ptr(), B, L, K, S, M, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            assert bool((guard[0] == 7.0).all()) and bool((guard[-1] == 7.0).all()), (pt, nt)
            assert torch.equal(q, out["0"]), (pt, nt)
    ref = _mix_ref(A.double().cpu(), Bt.double().cpu(), P.double().cpu(), Wp.double().cpu())
    assert rel_err(out["1"].cpu(), ref) < 1e-5


@pytest.mark.parametrize("B,L,K,S,M", [(2, 5, 3, 11, 24),          # small; wave 3 idle for L = 5? no: words 0..4 over 4 waves
                                       (2, 20, 10, 100, 1000),     # the training shape per sample (configs[3])
                                       (1, 24, 10, 7, 1024),       # six words per wave, full 1024 columns
                                       (2, 3, 10, 5, 8),           # fewer words than waves, one active lane pair
                                       (1, 21, 7, 3, 500),         # ragged word count, K < 10
                                       (1, 25, 10, 4, 64),         # L > 24: the LDS-weight kernel
                                       (2, 6, 12, 5, 40),          # K > 10: the LDS-weight kernel
                                       (2, 20, 10, 6, 4020),       # the folded layer-0 width: four column blocks
                                       (1, 5, 4, 3, 1028)])        # one column block + a 4-column one
def test_mix_bcast_add_bwd_matches_autograd(B, L, K, S, M, monkeypatch):
    """gA, gBt, gP, gWp of the concat-free mix vs float64 autograd of its einsum restatement; the
    register-weight pixel kernel (round 2) and the LDS-weight one must both pass and agree."""
    A, Bt = det_uniform((B, S, M), 111, 1.0), det_uniform((L, B, M), 112, 1.0)
    P, Wp = det_uniform((L, B, K, S), 113, 1.0), det_uniform((M, K), 114, 1.0)
    gQ = det_uniform((S * L, B, M), 115, 1.0)
    rs = [t.clone().double().requires_grad_(True) for t in (A, Bt, P, Wp)]
    (_mix_ref(*rs) * gQ.double()).sum().backward()
    got = {}
    monkeypatch.setenv("DMN_MIX_BWD_ONE", "0")                     # the two-pass kernels
    for mode in (("1", "0") if M <= 1024 else ("1", "1")):        # rows wider than 1024: round-2 kernels only
        monkeypatch.setenv("DMN_MIX_BWD_PIX2", mode)
        monkeypatch.setenv("DMN_MIX_BWD_WORD2", mode)
        gs = [t.to(DEV).requires_grad_(True) for t in (A, Bt, P, Wp)]
        (ops.mix_bcast_add(*gs) * gQ.to(DEV)).sum().backward()
        for a, b, n in zip(gs, rs, "A Bt P Wp".split()):
            assert rel_err(a.grad.cpu(), b.grad) < 1e-5, (mode, n)
        got[mode] = [a.grad.cpu() for a in gs]
    for a, b, n in zip(got["1"], got.get("0", got["1"]), "A Bt P Wp".split()):
        assert rel_err(a, b) < 2e-6, n
    if K <= 10 and L <= 20:
        # the one-pass kernel (round 3): same sums, gA / gBt in the same order as the two-pass kernels
        monkeypatch.setenv("DMN_MIX_BWD_ONE", "1")
        monkeypatch.setenv("DMN_MIX_BWD_PIX2", "1")
        monkeypatch.setenv("DMN_MIX_BWD_WORD2", "1")
        gs = [t.to(DEV).requires_grad_(True) for t in (A, Bt, P, Wp)]
        (ops.mix_bcast_add(*gs) * gQ.to(DEV)).sum().backward()
        for a, b, n in zip(gs, rs, "A Bt P Wp".split()):
            assert rel_err(a.grad.cpu(), b.grad) < 1e-5, ("one-pass", n)
        assert torch.equal(gs[0].grad.cpu(), got["1"][0]) and torch.equal(gs[1].grad.cpu(), got["1"][1])
        again = [t.to(DEV).requires_grad_(True) for t in (A, Bt, P, Wp)]
        (ops.mix_bcast_add(*again) * gQ.to(DEV)).sum().backward()
        for a, b in zip(gs, again):                                # deterministic: no atomics anywhere
            assert torch.equal(a.grad, b.grad)


@pytest.mark.parametrize("B,Cx,Cf,H,W", [(2, 3, 5, 4, 6), (1, 7, 0, 3, 3), (1, 0, 4, 1, 1), (2, 2, 2, 5, 7),
                                         (1, 33, 31, 16, 16)])
def test_cat_upsample2x_matches_torch(B, Cx, Cf, H, W):
    x, f = det_uniform((B, Cx, H, W), 121, 1.0), det_uniform((B, Cf, H, W), 122, 1.0)
    out = ops.cat_upsample2x(x.to(DEV), f.to(DEV)).cpu()
    ref = torch.nn.functional.interpolate(torch.cat([x, f], 1), scale_factor=2, mode="bilinear",
                                          align_corners=False)
    assert out.shape == ref.shape
    np.testing.assert_allclose(out.numpy(), ref.numpy(), rtol=1e-5, atol=1e-6)


def test_cat_upsample2x_bwd_matches_autograd():
    B, Cx, Cf, H, W = 2, 3, 2, 5, 6
    x, f = det_uniform((B, Cx, H, W), 123, 1.0), det_uniform((B, Cf, H, W), 124, 1.0)
    g = det_uniform((B, Cx + Cf, 2 * H, 2 * W), 125, 1.0)
    xg, fg = x.to(DEV).requires_grad_(True), f.to(DEV).requires_grad_(True)
    (ops.cat_upsample2x(xg, fg) * g.to(DEV)).sum().backward()
    xr, fr = x.clone().requires_grad_(True), f.clone().requires_grad_(True)
    (torch.nn.functional.interpolate(torch.cat([xr, fr], 1), scale_factor=2, mode="bilinear",
                                     align_corners=False) * g).sum().backward()
    np.testing.assert_allclose(xg.grad.cpu().numpy(), xr.grad.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(fg.grad.cpu().numpy(), fr.grad.numpy(), rtol=1e-5, atol=1e-6)


# ------------------------------------------------------------------------------------
# model level: the reference's own outputs (tests/golden, made by oracle/gen_golden.py)
# ------------------------------------------------------------------------------------
def _product_base(name):
    g = load_golden(name)
    cfg = cfg_from_golden(g)
    H, W, L, D = int(g["H"]), int(g["W"]), int(g["L"]), int(g["dict_size"])
    net = dms_amd.BaseDMN(D, emb_size=cfg["emb_size"], hid_size=cfg["hid_size"],
                          vis_size=cfg["vis_size"], num_filters=cfg["num_filters"],
                          mixed_size=cfg["mixed_size"], hid_mixed_size=cfg["hid_mixed_size"],
                          lang_layers=cfg["lang_layers"], mixed_layers=cfg["mixed_layers"],
                          mix_we=cfg["mix_we"], pretrained=False, high_res=cfg["high_res"])
    net.base = torch.nn.Identity()                      # golden vectors start after the backbone
    state = det_fill_state(net.state_dict())
    torch.nn.Module.load_state_dict(net, state)
    net = net.to(DEV).eval()
    vis, lang = base_inputs(cfg, H, W, L, D)
    return net, vis, lang, g, cfg, state


@pytest.mark.parametrize("name", ["base_tiny_k3", "base_tiny_k4", "base_tiny_k4_nowe_hr",
                                  "base_full_320_lowres", "base_full_512_highres_feat"])
def test_hot_path_matches_reference_golden(name):
    net, vis, lang, g, cfg, _ = _product_base(name)
    with torch.no_grad():
        out = net.hot_path(vis.to(DEV), lang.to(DEV)).cpu()
    assert tuple(out.shape) == g["out"].shape
    assert rel_err(out, g["out"]) < 1e-4                       # north_star tolerance (norm-wise)
    # ... and element by element: |a - b| <= 1e-4 |b| + 1e-6 max|b|.  One fixture needs a wider absolute
    # term: the (1, 1005, 16, 16) hidden-state output of the full-width mSRU (max |b| = 46).  Measured
    # here (float64 evaluation of oracle/dmn_ref on the same weights): the reference's OWN fp32 result is
    # 3.3e-3 = 7e-5 max|b| away from the exact value, so elements below ~1e-2 carry no fourth digit in
    # either implementation; the HIP path sits 1.6e-4 = 3.4e-6 max|b| from the reference, 20x closer.
    assert_close_elementwise(out, g["out"], atol_rel=1e-5 if name == "base_full_512_highres_feat" else 1e-6)
    if out.shape[1] == 1:                                      # thresholded mask agreement
        agree = ((out > 0) == (torch.from_numpy(g["out"]) > 0)).float().mean().item()
        assert agree >= 0.999


@pytest.mark.parametrize("B,L,K,S,d", [(2, 20, 10, 100, 1005), (1, 5, 3, 12, 7), (3, 4, 1, 7, 70), (2, 6, 10, 9, 64),
                                       (1, 3, 16, 5, 130), (2, 1, 10, 1, 5), (1, 20, 10, 3, 70), (3, 20, 10, 1, 64),
                                       # K = 10, L = 20 with B*d a multiple of 16: the two-wave kernel with its
                                       # 64-byte-aligned (shifted) column windows, incl. windows that start below column 0
                                       (16, 20, 10, 3, 1005), (32, 20, 10, 2, 1005), (2, 20, 10, 4, 1000), (4, 20, 10, 2, 100),
                                       # ... the matrix-core kernel (blocks of 32 steps, super-blocks of 8 pixels): one and
                                       # several whole super-blocks, ragged last window, and partial last super-blocks
                                       # (S % 8 = 1 .. 7: the end falls in every block and on different 4-step groups)
                                       (2, 20, 10, 8, 64), (1, 20, 10, 16, 1005), (3, 20, 10, 24, 1000), (16, 20, 10, 8, 1005),
                                       (5, 20, 10, 8, 70), (2, 20, 10, 9, 64), (1, 20, 10, 13, 130), (2, 20, 10, 18, 70),
                                       (1, 20, 10, 5, 64), (2, 20, 10, 6, 100), (1, 20, 10, 7, 64), (1, 20, 10, 23, 64)])
@pytest.mark.parametrize("act", ACTS)
def test_mix_fused_into_layer0_scan_is_bit_identical_to_mix_then_scan(B, L, K, S, d, act):
    """dmn_mix_sru_fwd == dmn_sru_fwd(dmn_mix_bcast_add(...)) bit for bit (h and c_last), with and
    without an initial state, and both match the CPU oracle on the einsum-built U0."""
    M4 = 4 * d
    A = det_uniform((B, S, M4), 201, 1.0).to(DEV)
    Bt = det_uniform((L, B, M4), 202, 1.0).to(DEV)
    P = det_uniform((L, B, K, S), 203, 1.0).to(DEV)
    Wfull = det_uniform((M4, K + 3), 204, 0.5).to(DEV)
    Wp = Wfull[:, 2:2 + K]                                     # a column slice: ldw > K
    bias = det_uniform((2 * d,), 205, 0.5).to(DEV)
    c0 = det_uniform((B, d), 206, 1.0).to(DEV)
    for init in (None, c0):
        U0 = ops.mix_bcast_add(A, Bt, P, Wp).view(S * L, B, d, 4)
        h_ref, _, c_ref = ops.sru_forward_raw(U0, None, bias, init, act, 1.0, nchunks=1)
        h, c = ops.mix_sru_fwd(A, Bt, P, Wp, bias, init, act)
        assert h.shape == (S * L, B, d) and torch.equal(h, h_ref) and torch.equal(c, c_ref)
        if K == 10 and L == 20:                                # both two-wave kernels, whatever the library picks
            for force in ("0", "1"):
                os.environ["DMN_MIX_SRU_MFMA"] = force
                try:
                    hv, cv = ops.mix_sru_fwd(A, Bt, P, Wp, bias, init, act)
                finally:
                    del os.environ["DMN_MIX_SRU_MFMA"]
                assert torch.equal(hv, h_ref) and torch.equal(cv, c_ref), "DMN_MIX_SRU_MFMA=" + force
    want = (A.permute(1, 0, 2)[:, None] + Bt[None] + torch.einsum("mk,tbks->stbm", Wp, P)).reshape(S * L, B, d, 4)
    h_o, _, c_o = sru_ref.sru_fwd_c(want.cpu().numpy(), None, bias.cpu().numpy(), c0.cpu().numpy(), act, 1.0)
    np.testing.assert_allclose(h.cpu().numpy(), h_o, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(c.cpu().numpy(), c_o, rtol=1e-4, atol=1e-5)


def test_hot_path_with_fused_layer0_at_batch_16_equals_two_kernel_path_and_batch1():
    """Full dims (d = 1005, 20 words, 10 filters), 16 samples: above MIX_SCAN_MIN_COLS the hot path
    takes dmn_mix_sru_fwd by itself; it must equal the mix + scan path bit for bit (both scans are
    un-chunked at this size) and every sample must equal its own batch-1 forward (reference
    semantics, dmn.py:142)."""
    net, vis, lang, g, cfg, _ = _product_base("base_full_320_lowres")
    import dms_amd.models.dmn as dmn_mod
    B = 16
    assert B * net.mrnn.rnn_lst[0].n_out >= dmn_mod.MIX_SCAN_MIN_COLS
    gen = torch.Generator().manual_seed(3)
    visb = torch.cat([vis] + [torch.rand(vis.shape, generator=gen) for _ in range(B - 1)]).to(DEV)
    langb = torch.cat([lang] + [torch.randint(0, int(g["dict_size"]), lang.shape, generator=gen)
                                for _ in range(B - 1)]).to(DEV)
    with torch.no_grad():
        fused = net.hot_path(visb, langb)
        net.fuse_mix_scan = False
        pair = net.hot_path(visb, langb)
        one = net.hot_path(visb[5:6], langb[5:6])
    assert torch.equal(fused, pair)
    assert rel_err(fused[:1].cpu(), g["out"]) < 1e-4
    assert rel_err(fused[5:6], one) < 1e-5


def test_folded_layer0_projection_equals_unfused_path():
    """comb_conv folded into the mSRU layer-0 projection (inference) vs the unfused path."""
    net, vis, lang, g, cfg, _ = _product_base("base_full_320_lowres")
    with torch.no_grad():
        assert net.fold_comb_into_msru and net.mrnn.rnn_lst[0].k == 4
        a = net.hot_path(vis.to(DEV), lang.to(DEV)).cpu()
        # the same with the mix fused into the layer-0 scan (taken by itself from 6 samples on)
        import dms_amd.models.dmn as dmn_mod
        keep = dmn_mod.MIX_SCAN_MIN_COLS
        dmn_mod.MIX_SCAN_MIN_COLS = 0
        try:
            a2 = net.hot_path(vis.to(DEV), lang.to(DEV)).cpu()
        finally:
            dmn_mod.MIX_SCAN_MIN_COLS = keep
        assert rel_err(a2, a) < 1e-5       # (bit-identical only vs the un-chunked scan; B=1 runs the time-parallel one)
        net.fold_comb_into_msru = False
        b = net.hot_path(vis.to(DEV), lang.to(DEV)).cpu()
        # cache follows in-place weight updates
        net.fold_comb_into_msru = True
        net.comb_conv.weight.mul_(0.5)
        c = net.hot_path(vis.to(DEV), lang.to(DEV)).cpu()
        net.fold_comb_into_msru = False
        d = net.hot_path(vis.to(DEV), lang.to(DEV)).cpu()
    assert rel_err(a, b) < 2e-5 and rel_err(a, g["out"]) < 1e-4 and rel_err(b, g["out"]) < 1e-4
    assert rel_err(c, d) < 2e-5 and rel_err(c, a) > 1e-3


def test_hot_path_batched_equals_independent_batch1_forwards():
    net, vis, lang, g, cfg, state = _product_base("base_tiny_k4")
    B = 3
    visb = torch.cat([vis, det_uniform(tuple(vis.shape), 131, 1.0), det_uniform(tuple(vis.shape), 132, 1.0)])
    langb = torch.cat([lang, det_ints(tuple(lang.shape), 133, 50), det_ints(tuple(lang.shape), 134, 50)])
    with torch.no_grad():
        out = net.hot_path(visb.to(DEV), langb.to(DEV)).cpu()
        P = {"langvis." + k: v for k, v in state.items()}
        ref = dmn_ref.hot_path(P, visb, langb, cfg)
    assert out.shape == ref.shape == (B, 1, 4, 3)
    assert rel_err(out, ref) < 1e-4
    assert rel_err(out[:1], g["out"]) < 1e-4


def test_upsampling_module_matches_reference_golden():
    g = load_golden("upsample_small")
    fch = [6, 5, 4, 3, 2]
    up = dms_amd.UpsamplingModule(7, mode="bilinear", ker_size=3, amplification=8, feature_channels=fch)
    up.load_state_dict(det_fill_state(up.state_dict()))
    up = up.to(DEV).eval()
    x = det_uniform((1, 7, 3, 4), 303, 1.0).to(DEV)
    feats = [None, None, det_uniform((1, fch[2], 12, 16), 306, 1.0).to(DEV),
             det_uniform((1, fch[1], 6, 8), 305, 1.0).to(DEV), det_uniform((1, fch[0], 3, 4), 304, 1.0).to(DEV)]
    with torch.no_grad():
        y = up(x, feats).cpu()
    assert rel_err(y, g["out"]) < 1e-4
    assert_close_elementwise(y, g["out"])
    feats = [None, None, det_uniform((1, fch[2], 13, 15), 309, 1.0).to(DEV),
             det_uniform((1, fch[1], 7, 9), 308, 1.0).to(DEV), det_uniform((1, fch[0], 4, 5), 307, 1.0).to(DEV)]
    with torch.no_grad():
        y = up(x, feats).cpu()
    assert rel_err(y, g["out_mis"]) < 1e-4


@pytest.mark.parametrize("B,Cx,Cf,H,W,Cout", [(2, 3, 5, 4, 8, 6), (1, 2, 3, 3, 5, 4), (1, 4, 0, 2, 4, 3),
                                              (2, 16, 16, 16, 16, 32), (1, 3, 2, 1, 1, 2), (1, 2, 2, 1, 4, 2),
                                              (3, 5, 4, 7, 12, 9), (1, 3, 3, 2, 2, 2), (2, 4, 3, 6, 10, 5),
                                              (1, 8, 8, 8, 24, 4)])
@pytest.mark.parametrize("m", [2, 4])
def test_winograd_level_matches_cat_upsample_conv(B, Cx, Cf, H, W, Cout, m):
    """transform -> (m+2)^2 GEMMs -> transform == conv3x3(upsample2x(cat)) (dmn.py:324,337-338)."""
    x = det_uniform((B, Cx, H, W), 81, 1.0).to(DEV)
    f = det_uniform((B, Cf, H, W), 82, 1.0).to(DEV)
    if m == 4 and (H % 2 or W % 2):
        with pytest.raises(RuntimeError, match="even"):
            ops.wino_cat_upsample2x_transform(x, f, 4)
        return
    w = det_uniform((Cout, Cx + Cf, 3, 3), 83, 0.3).to(DEV)
    bias = det_uniform((Cout,), 84, 0.5).to(DEV)
    up = F.interpolate(torch.cat([x, f], 1), scale_factor=2, mode="bilinear", align_corners=False)
    want = F.conv2d(up.double(), w.double(), bias.double(), padding=1).float()
    V = ops.wino_cat_upsample2x_transform(x, f, m)
    assert V.shape == ((m + 2) ** 2, Cx + Cf, B * (2 * H // m) * (2 * W // m))
    M = torch.bmm(ops.wino_filter_transform(w, m), V)
    got = ops.wino_output_transform(M, bias, B, H, W, m)
    tol = 1e-5 if m == 2 else 5e-5                     # F(4x4) trades ~15x of fp32 accuracy for speed
    assert got.shape == want.shape and rel_err(got, want) < tol
    got_nobias = ops.wino_output_transform(M, None, B, H, W, m)
    assert rel_err(got_nobias + bias.view(1, -1, 1, 1), want) < tol


@pytest.mark.parametrize("B,Cx,Cf,H,W,Co", [(2, 3, 5, 4, 8, 6), (1, 2, 3, 3, 5, 4), (1, 4, 0, 1, 1, 3), (2, 8, 8, 16, 16, 32),
                                            (1, 2, 2, 7, 12, 5)])
def test_head_final_equals_conv_then_out_layer(B, Cx, Cf, H, W, Co):
    """conv3x3(C -> Co) . out_layer(Co -> 1) composed into one C -> 1 conv on the upsampled cat."""
    x = det_uniform((B, Cx, H, W), 131, 1.0).to(DEV)
    f = det_uniform((B, Cf, H, W), 132, 1.0).to(DEV)
    w = det_uniform((Co, Cx + Cf, 3, 3), 133, 0.3).to(DEV)
    bias = det_uniform((Co,), 134, 0.5).to(DEV)
    wo = det_uniform((1, Co, 1, 1), 135, 0.5).to(DEV)
    bo = det_uniform((1,), 136, 0.5).to(DEV)
    up = F.interpolate(torch.cat([x, f], 1), scale_factor=2, mode="bilinear", align_corners=False)
    want = F.conv2d(F.conv2d(up.double(), w.double(), bias.double(), padding=1), wo.double(), bo.double()).float()
    weff = torch.einsum("o,ocyx->cyx", wo.view(-1), w).contiguous()
    beff = (bo + (wo.view(-1) * bias).sum()).view(1)
    got = ops.head_final(x, f, weff, beff)
    assert got.shape == want.shape and rel_err(got, want) < 1e-5


@pytest.mark.parametrize("B,Cx,Cf,H,W,Cout", [(2, 3, 5, 4, 8, 6), (1, 2, 3, 3, 5, 4), (1, 4, 0, 2, 4, 3),
                                              (2, 16, 16, 16, 16, 32), (1, 3, 2, 1, 1, 2), (1, 2, 2, 1, 4, 2),
                                              (3, 5, 4, 7, 12, 9), (1, 3, 3, 2, 2, 2), (2, 4, 3, 6, 10, 5),
                                              (1, 8, 8, 8, 24, 4), (1, 0, 3, 5, 3, 2), (2, 3, 3, 40, 20, 3),
                                              (1, 2, 1, 33, 130, 2),
                                              # power-of-two widths: the tiled kernel -- whole planes per band
                                              # (H <= 1024 / W), bands with halo rows (H > 1024 / W), non-square
                                              (1, 3, 2, 32, 32, 4), (1, 2, 2, 64, 64, 2), (1, 1, 1, 128, 128, 2),
                                              (2, 2, 1, 8, 32, 4), (1, 2, 2, 32, 64, 3), (1, 2, 1, 128, 16, 2),
                                              (2, 3, 1, 16, 128, 3),
                                              # W = 256: bands of 4 rows, one row per wavefront (the last head level)
                                              (1, 2, 1, 256, 256, 1), (2, 1, 2, 8, 256, 3), (1, 1, 1, 4, 256, 2)])
def test_tap_gemm_level_matches_cat_upsample_conv(B, Cx, Cf, H, W, Cout, monkeypatch):
    """nine tap matrices at low resolution (library GEMM) + gather at high resolution ==
    conv3x3(upsample2x(cat)) (dmn.py:324,337-338), incl. odd sizes, 1-pixel maps, empty sources."""
    x = det_uniform((B, Cx, H, W), 91, 1.0).to(DEV)
    f = det_uniform((B, Cf, H, W), 92, 1.0).to(DEV)
    w = det_uniform((Cout, Cx + Cf, 3, 3), 93, 0.3).to(DEV)
    bias = det_uniform((Cout,), 94, 0.5).to(DEV)
    up = F.interpolate(torch.cat([x, f], 1), scale_factor=2, mode="bilinear", align_corners=False)
    want = F.conv2d(up.double(), w.double(), bias.double(), padding=1).float()
    wt = ops.tap_weight(w)
    assert wt.shape == (9 * Cout, Cx + Cf) and torch.equal(wt[4 * Cout:5 * Cout], w[:, :, 1, 1])
    Z = ops.tap_gemm(x, f, wt[:, :Cx].contiguous(), wt[:, Cx:].contiguous())
    assert Z.shape == (B, 9, Cout, H, W)
    got = ops.tap_upsample_gather(Z, bias)
    assert got.shape == want.shape and rel_err(got, want) < 2e-6
    got_nobias = ops.tap_upsample_gather(Z, None)
    assert rel_err(got_nobias + bias.view(1, -1, 1, 1), want) < 2e-6
    # the row-sweep kernel (all shapes) and the tiled kernel (power-of-two widths) agree to rounding,
    # also when the output is a channel slice of a larger buffer (out_bs > Cout * 4 * H * W)
    monkeypatch.setenv("DMN_TAP_TILE", "0")
    sweep = ops.tap_upsample_gather(Z, bias)
    monkeypatch.delenv("DMN_TAP_TILE")
    assert rel_err(got, sweep) < 1e-6
    big = torch.full((B, Cout + 3, 2 * H, 2 * W), 7.0, device=DEV)
    ops.tap_upsample_gather(Z, bias, out=big[:, :Cout])
    assert torch.equal(big[:, :Cout], got) and bool((big[:, Cout:] == 7.0).all())


@pytest.mark.parametrize("B,Cx,Cf,H,W,Co", [(2, 3, 5, 4, 8, 6), (1, 2, 3, 3, 5, 4), (1, 4, 0, 1, 1, 3), (2, 8, 8, 16, 16, 32),
                                            (1, 2, 2, 7, 12, 5), (1, 0, 6, 9, 7, 2), (2, 64, 64, 32, 64, 32)])
def test_tap_project_and_gather_equal_conv_then_out_layer(B, Cx, Cf, H, W, Co):
    """last level composed with out_layer: 9-row projection kernel + gather == conv3x3 . out_layer."""
    x = det_uniform((B, Cx, H, W), 131, 1.0).to(DEV)
    f = det_uniform((B, Cf, H, W), 132, 1.0).to(DEV)
    w = det_uniform((Co, Cx + Cf, 3, 3), 133, 0.3).to(DEV)
    bias = det_uniform((Co,), 134, 0.5).to(DEV)
    wo = det_uniform((1, Co, 1, 1), 135, 0.5).to(DEV)
    bo = det_uniform((1,), 136, 0.5).to(DEV)
    up = F.interpolate(torch.cat([x, f], 1), scale_factor=2, mode="bilinear", align_corners=False)
    want = F.conv2d(F.conv2d(up.double(), w.double(), bias.double(), padding=1), wo.double(), bo.double()).float()
    weff = torch.einsum("o,ocyx->cyx", wo.view(-1), w).contiguous()
    beff = (bo + (wo.view(-1) * bias).sum()).view(1)
    w9 = weff.view(Cx + Cf, 9).t().contiguous()
    Z = ops.tap_project(x, f, w9)
    assert Z.shape == (B, 9, 1, H, W)
    zref = torch.einsum("rc,bchw->brhw", w9.double(), torch.cat([x, f], 1).double()).float()
    assert rel_err(Z.view(B, 9, H, W), zref) < 2e-6
    got = ops.tap_upsample_gather(Z, beff)
    assert got.shape == want.shape and rel_err(got, want) < 5e-6


def test_upsampling_module_winograd_path_matches_reference_golden():
    g = load_golden("upsample_small")
    fch = [6, 5, 4, 3, 2]
    up = dms_amd.UpsamplingModule(7, mode="bilinear", ker_size=3, amplification=8, feature_channels=fch)
    up.load_state_dict(det_fill_state(up.state_dict()))
    up = up.to(DEV).eval()
    x = det_uniform((1, 7, 3, 4), 303, 1.0).to(DEV)
    feats = [None, None, det_uniform((1, fch[2], 12, 16), 306, 1.0).to(DEV),
             det_uniform((1, fch[1], 6, 8), 305, 1.0).to(DEV), det_uniform((1, fch[0], 3, 4), 304, 1.0).to(DEV)]
    with torch.no_grad():
        y_tap = up(x, feats).cpu()                            # default: tap GEMM + gather levels
        up.tap_gemm = False
        up.winograd = False
        y_plain = up(x, feats).cpu()
        assert rel_err(y_tap, g["out"]) < 1e-4 and rel_err(y_tap, y_plain) < 1e-5
        up.fuse_last = False
        assert rel_err(up(x, feats).cpu(), y_plain) < 1e-5      # conv + out_layer unfused
        up.fuse_last = True
        up.winograd = "f2"
        y = up(x, feats).cpu()
        assert rel_err(y, g["out"]) < 1e-4 and rel_err(y, y_plain) < 1e-5
        up.winograd = True                                    # F(4x4) on the even levels, F(2x2) on 3x4
        y = up(x, feats).cpu()
        assert rel_err(y, g["out"]) < 1e-4 and rel_err(y, y_plain) < 1e-4
    up.winograd = True
    feats = [None, None, det_uniform((1, fch[2], 13, 15), 309, 1.0).to(DEV),
             det_uniform((1, fch[1], 7, 9), 308, 1.0).to(DEV), det_uniform((1, fch[0], 4, 5), 307, 1.0).to(DEV)]
    with torch.no_grad():
        y = up(x, feats).cpu()                                 # resize branch + odd widths (scalar kernels)
    assert rel_err(y, g["out_mis"]) < 1e-4
    # with autograd the plain path runs
    y_g = up(x, feats)
    assert y_g.requires_grad


def test_whole_dmn_winograd_head_matches_reference_end_to_end():
    g = load_golden("dmn_e2e_64px_highres")
    net = dms_amd.DMN(100, num_filters=10, lang_layers=3, mix_we=True, pretrained=False,
                      high_res=True, upsampling_mode="bilinear", upsampling_amplification=32)
    torch.nn.Module.load_state_dict(net, det_fill_state(net.state_dict()))
    net = net.to(DEV).eval()
    net.upsample.tap_gemm = False
    net.upsample.winograd = True
    img = det_uniform((1, 3, 64, 64), 404, 1.0).to(DEV)
    words = det_ints((1, 6), 405, 100).to(DEV)
    with torch.no_grad():
        y = net(img, words).cpu()
    assert rel_err(y, g["out"]) < 1e-4
    # optional F(4x4) Winograd head (not the default path): ~1e-5 of max|level output| per level
    assert_close_elementwise(y, g["out"], rtol=2e-4, atol_rel=1e-5)


def test_whole_dmn_matches_reference_end_to_end():
    """Reference DMN incl. its own DPN92 on a 64 px image, high-res head (golden e2e)."""
    g = load_golden("dmn_e2e_64px_highres")
    net = dms_amd.DMN(100, num_filters=10, lang_layers=3, mix_we=True, pretrained=False,
                      high_res=True, upsampling_mode="bilinear", upsampling_amplification=32)
    torch.nn.Module.load_state_dict(net, det_fill_state(net.state_dict()))
    net = net.to(DEV).eval()
    img = det_uniform((1, 3, 64, 64), 404, 1.0).to(DEV)
    words = det_ints((1, 6), 405, 100).to(DEV)
    with torch.no_grad():
        vis, feats = net.langvis.base(img)
        y = net(img, words).cpu()
    assert rel_err(vis.cpu(), g["vis"]) < 1e-4                 # backbone (MIOpen) vs reference CPU
    assert rel_err(feats[3].cpu(), g["feat3"]) < 1e-4
    assert y.shape == (1, 1, 64, 64)
    assert rel_err(y, g["out"]) < 1e-4
    # whole network incl. ~100 MIOpen / rocBLAS convolutions vs the reference on torch-CPU: the backbone
    # features themselves differ by ~1e-6 max|.| (asserted above to 1e-4), so the absolute term is 1e-5 max|ref|
    assert_close_elementwise(y, g["out"], atol_rel=1e-5)
    agree = ((y > 0) == (torch.from_numpy(g["out"]) > 0)).float().mean().item()
    assert agree >= 0.999


# ------------------------------------------------------------------------------------
# DPN backbone glue (inference): BN+ReLU and the dual-path merge
# ------------------------------------------------------------------------------------
@pytest.mark.parametrize("N,C,H,W", [(2, 5, 4, 4), (1, 3, 3, 5), (3, 64, 8, 8), (1, 1, 1, 1)])
def test_bn_relu_matches_torch_eval_batchnorm(N, C, H, W):
    bn = torch.nn.BatchNorm2d(C, eps=1e-3)
    bn.load_state_dict(det_fill_state(bn.state_dict()))
    bn.running_var.abs_().add_(0.1)
    bn = bn.to(DEV).eval()
    x = det_uniform((N, C, H, W), 71, 2.0).to(DEV)
    scale = bn.weight.detach() * torch.rsqrt(bn.running_var + bn.eps)
    got = ops.bn_relu(x, bn.running_mean, scale.contiguous(), bn.bias.detach())
    with torch.no_grad():
        want = torch.relu(bn(x))
    assert rel_err(got, want) < 1e-6


@pytest.mark.parametrize("N,C,H,W", [(2, 5, 4, 4), (1, 3, 7, 12), (2, 64, 32, 32), (1, 1, 1, 4), (1, 2, 9, 8), (2, 3, 160, 160)])
def test_bn_relu_maxpool_matches_torch_stem(N, C, H, W):
    """stem bn + relu + MaxPool2d(3, 2, 1) (dpn_model.py:186-197) in one pass; bit-exact pooling."""
    x = det_uniform((N, C, H, W), 181, 2.0).to(DEV)
    mean = det_uniform((C,), 182, 0.5).to(DEV)
    scale = det_uniform((C,), 183, 1.5).to(DEV)               # negative scales too
    bias = det_uniform((C,), 184, 0.5).to(DEV)
    want = F.max_pool2d(ops.bn_relu(x, mean, scale, bias), 3, 2, 1)
    got = ops.bn_relu_maxpool(x, mean, scale, bias)
    assert got.shape == want.shape and torch.equal(got, want)


@pytest.mark.parametrize("N,C,H,W", [(2, 5, 4, 4), (1, 3, 6, 12), (2, 64, 8, 8), (1, 1, 2, 4)])
def test_bn_relu_stride2_matches_subsampled_torch(N, C, H, W):
    x = det_uniform((N, C, H, W), 111, 2.0).to(DEV)
    bn = (det_uniform((C,), 112, 1.0).to(DEV), det_uniform((C,), 113, 1.0).to(DEV).abs() + 0.5,
          det_uniform((C,), 114, 0.5).to(DEV))
    got = ops.bn_relu_s2(x, *bn)
    assert torch.equal(got, ops.bn_relu(x, *bn)[:, :, ::2, ::2].contiguous())
    with pytest.raises(RuntimeError, match="needs H"):
        ops.bn_relu_s2(x[:, :, :, :W - 2].contiguous(), *bn)


@pytest.mark.parametrize("N,C,H,W", [(2, 32, 8, 8), (1, 1005, 10, 10), (3, 7, 3, 5), (1, 1, 1, 1), (2, 32, 64, 64)])
def test_channel_dot_matches_1x1_conv_to_one_channel(N, C, H, W):
    x = det_uniform((N, C, H, W), 121, 1.0).to(DEV)
    w = det_uniform((1, C, 1, 1), 122, 0.3).to(DEV)
    b = det_uniform((1,), 123, 0.5).to(DEV)
    want = F.conv2d(x.double(), w.double(), b.double()).float()
    assert rel_err(ops.channel_dot(x, w.view(-1), b), want) < 1e-5
    assert rel_err(ops.channel_dot(x, w.view(-1), None), want - b.view(1, 1, 1, 1)) < 1e-5


@pytest.mark.parametrize("N,Cb,Cy,c1c,H,W", [(2, 7, 5, 3, 4, 4), (1, 6, 4, 4, 3, 5), (2, 336, 272, 256, 8, 8),
                                             (1, 4, 4, 4, 2, 2)])
@pytest.mark.parametrize("with_bn", [False, True])
def test_dpn_merge_matches_add_and_cat(N, Cb, Cy, c1c, H, W, with_bn):
    base = det_uniform((N, Cb, H, W), 72, 1.0).to(DEV)
    y = det_uniform((N, Cy, H, W), 73, 1.0).to(DEV)
    want = torch.cat([base[:, :c1c] + y[:, :c1c], base[:, c1c:], y[:, c1c:]], 1)
    Co = Cb + Cy - c1c
    bn = None
    if with_bn:
        bn = (det_uniform((Co,), 74, 1.0).to(DEV), det_uniform((Co,), 75, 1.0).to(DEV).abs() + 0.5,
              det_uniform((Co,), 76, 1.0).to(DEV))
    out, act = ops.dpn_merge(base, y, c1c, bn)
    assert torch.equal(out, want)                               # add + copy: bit exact
    if with_bn:
        m, s, b = (t.view(1, -1, 1, 1) for t in bn)
        assert rel_err(act, torch.relu((want - m) * s + b)) < 1e-6
    else:
        assert act is None


@pytest.mark.parametrize("N,groups,cg,H,W", [(2, 4, 3, 8, 8), (1, 2, 6, 5, 7), (2, 32, 3, 16, 64), (1, 32, 12, 32, 32),
                                             (2, 32, 24, 16, 16), (1, 3, 4, 9, 12), (1, 2, 8, 3, 3), (1, 2, 16, 6, 20),
                                             (1, 1, 32, 4, 4), (1, 2, 3, 1, 1), (1, 4, 6, 20, 130)])
@pytest.mark.parametrize("stages", ["both", "none", "in", "out"])
def test_grouped_conv3x3_with_fused_bn_relu_matches_torch(N, groups, cg, H, W, stages):
    C = groups * cg
    x = det_uniform((N, C, H, W), 91, 2.0).to(DEV)
    w = det_uniform((C, cg, 3, 3), 92, 0.5).to(DEV)

    def consts(seed):
        return (det_uniform((C,), seed, 1.0).to(DEV), det_uniform((C,), seed + 1, 1.0).to(DEV).abs() + 0.5,
                det_uniform((C,), seed + 2, 0.5).to(DEV))

    bn_in = consts(93) if stages in ("both", "in") else None
    bn_out = consts(96) if stages in ("both", "out") else None

    def bn_relu(t, bn):
        m, sc, b = (v.view(1, -1, 1, 1).double() for v in bn)
        return torch.relu((t - m) * sc + b)

    t = x.double()
    if bn_in is not None:
        t = bn_relu(t, bn_in)
    t = F.conv2d(t, w.double(), None, 1, 1, 1, groups)
    if bn_out is not None:
        t = bn_relu(t, bn_out)
    got = ops.gconv3x3_bn_relu(x, ops.gconv_weight_layout(w, groups), groups, bn_in, bn_out)
    assert got.shape == x.shape and rel_err(got, t.float()) < 1e-5


@pytest.mark.parametrize("N,groups,cg,H,W", [(2, 4, 3, 8, 8), (1, 32, 6, 16, 64), (2, 32, 12, 32, 32), (1, 8, 24, 32, 32),
                                             (1, 2, 4, 6, 24), (1, 2, 16, 2, 8), (1, 2, 6, 20, 272)])
def test_grouped_conv3x3_stride2_matches_torch(N, groups, cg, H, W):
    C = groups * cg
    x = det_uniform((N, C, H, W), 101, 2.0).to(DEV)
    w = det_uniform((C, cg, 3, 3), 102, 0.5).to(DEV)
    bn1 = (det_uniform((C,), 103, 1.0).to(DEV), det_uniform((C,), 104, 1.0).to(DEV).abs() + 0.5, det_uniform((C,), 105, 0.5).to(DEV))
    bn2 = (det_uniform((C,), 106, 1.0).to(DEV), det_uniform((C,), 107, 1.0).to(DEV).abs() + 0.5, det_uniform((C,), 108, 0.5).to(DEV))

    def bn_relu(t, bn):
        m, sc, b = (v.view(1, -1, 1, 1).double() for v in bn)
        return torch.relu((t - m) * sc + b)

    want = bn_relu(F.conv2d(bn_relu(x.double(), bn1), w.double(), None, 2, 1, 1, groups), bn2).float()
    got = ops.gconv3x3_bn_relu(x, ops.gconv_weight_layout(w, groups), groups, bn1, bn2, stride=2)
    assert got.shape == want.shape and rel_err(got, want) < 1e-5
    with pytest.raises(RuntimeError, match="stride 2 needs"):
        ops.gconv3x3_bn_relu(x[:, :, :, :W - 4].contiguous(), ops.gconv_weight_layout(w, groups), groups, bn1, bn2, stride=2)


def test_grouped_conv3x3_rejects_unsupported_group_width():
    x = torch.zeros((1, 10, 4, 4), device=DEV)
    w = torch.zeros((2, 5, 9, 5), device=DEV)
    with pytest.raises(RuntimeError, match="not instantiated"):
        ops.gconv3x3_bn_relu(x, w, 2)


@pytest.mark.parametrize("N,c1c,inc,k,H,W", [(2, 8, 2, 0, 4, 4), (1, 6, 3, 2, 3, 5), (2, 256, 16, 1, 8, 8), (1, 4, 1, 5, 1, 1)])
@pytest.mark.parametrize("want_out", [False, True])
def test_dpn_stage_pass_matches_reference_order_and_moves_the_slot(N, c1c, inc, k, H, W, want_out):
    """In-place stage buffer [dense0 | res | slot | o2_1..]: the pass must emit relu(bn(X)) with X in
    the reference's order [res | dense0 | o2_1..o2_k | slot], move the slot to its final rows and
    zero it (dpn_model.py:246,265-269)."""
    Co = c1c + 2 * inc + (k + 1) * inc
    rows = Co + inc + 2 * inc                                  # room for two more blocks
    buf = det_uniform((N, rows, H, W), 191, 1.0).to(DEV)
    before = buf.clone()
    mean, scale, bias = (det_uniform((Co,), 192 + i, 1.0).to(DEV) for i in range(3))
    dense0, res = before[:, :2 * inc], before[:, 2 * inc:2 * inc + c1c]
    slot = before[:, 2 * inc + c1c:3 * inc + c1c]
    older = before[:, 3 * inc + c1c:3 * inc + c1c + k * inc]
    X = torch.cat([res, dense0, older, slot], 1)
    assert X.shape[1] == Co
    act, out = ops.dpn_stage_pass(buf, c1c, inc, Co, (mean, scale, bias), want_out)
    want = torch.relu((X - mean.view(1, -1, 1, 1)) * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1))
    assert rel_err(act, want) < 1e-6
    assert (out is None) == (not want_out) and (out is None or torch.equal(out, X))
    assert torch.equal(buf[:, Co:Co + inc], slot)                              # moved to its final rows
    assert not buf[:, 2 * inc + c1c:3 * inc + c1c].any()                        # slot cleared
    keep = torch.ones(rows, dtype=torch.bool)
    keep[2 * inc + c1c:3 * inc + c1c] = False
    keep[Co:Co + inc] = False
    assert torch.equal(buf[:, keep], before[:, keep])                           # nothing else touched


def test_dpn92_fused_glue_equals_plain_torch_backbone():
    """Same module, same weights: fast path (HIP glue) vs the plain torch path, all taps."""
    from dms_amd.models.dpn import DPN
    net = DPN("dpn92")
    net.load_state_dict(det_fill_state(net.state_dict()))
    net = net.to(DEV).eval()
    img = det_uniform((2, 3, 96, 64), 77, 1.0).to(DEV)
    with torch.no_grad():
        out_f, feats_f = net(img)
        net.fuse_glue = False
        out_t, feats_t = net(img)
    assert len(feats_f) == len(feats_t) == 5
    assert rel_err(out_f, out_t) < 1e-5
    for a, b in zip(feats_f, feats_t):
        assert a.shape == b.shape and rel_err(a, b) < 1e-5
    # per-block dpn_merge instead of the in-place stage buffers; odd sizes take it by themselves
    net.fuse_glue, net.stage_inplace = True, False
    with torch.no_grad():
        out_m, feats_m = net(img)
        net.stage_inplace = True
        odd = det_uniform((1, 3, 72, 88), 78, 1.0).to(DEV)     # 18x22 maps at conv2: W % 4 != 0 before conv3
        out_o, feats_o = net(odd)
        net.fuse_glue = False
        out_ot, feats_ot = net(odd)
    assert rel_err(out_m, out_t) < 1e-5 and all(rel_err(a, b) < 1e-5 for a, b in zip(feats_m, feats_t))
    assert rel_err(out_o, out_ot) < 1e-5 and all(rel_err(a, b) < 1e-5 for a, b in zip(feats_o, feats_ot))
    # weights changed in place (load_state_dict) must invalidate the cached BN constants
    net.fuse_glue = True
    with torch.no_grad():
        net.features.conv5_bn_ac.bn.running_mean.add_(0.25)
        out_f2, _ = net(img)
        net.fuse_glue = False
        out_t2, _ = net(img)
    assert rel_err(out_f2, out_t2) < 1e-5 and not torch.equal(out_f2, out_f)
    # training mode / grad enabled keeps the autograd-capable torch path
    net.fuse_glue = True
    out_g, _ = net(img)
    assert out_g.requires_grad


@pytest.mark.parametrize("high_res,size", [(True, (320, 320)), (False, (320, 320)), (True, (224, 352))])
def test_dmn_every_inference_shortcut_equals_the_plain_path(high_res, size):
    """BASELINE configs[0]/[3] image size (320 px: 10x10 .. 160x160 maps, widths that are not
    multiples of 4 or 8) and a non-square one: the forward with every inference shortcut on
    (DPN glue + grouped conv kernels, comb_conv fold, state-only scan, Winograd levels, fused
    last level, tap-GEMM head levels, channel reduction, per-shape BLAS) must equal the forward with all of them off
    (torch modules + MIOpen + the unfused HIP path)."""
    net = dms_amd.DMN(100, num_filters=10, lang_layers=3, mix_we=True, pretrained=False,
                      high_res=high_res, upsampling_mode="bilinear", upsampling_amplification=32)
    torch.nn.Module.load_state_dict(net, det_fill_state(net.state_dict()))
    net = net.to(DEV).eval()
    img = det_uniform((2, 3) + size, 141, 1.0).to(DEV)
    words = det_ints((2, 7), 142, 100).to(DEV)
    with torch.no_grad():
        fast = net(img, words)
        net.langvis.base.fuse_glue = False
        net.langvis.fold_comb_into_msru = False
        for cell in net.langvis.mrnn.rnn_lst:
            cell.state_only_head = False
        if high_res:
            net.upsample.tap_gemm = False
            net.upsample.winograd = False
            net.upsample.fuse_last = False
        plain = net(img, words)
    assert fast.shape == plain.shape == ((2, 1) + size if high_res else (2, 1, size[0] // 32, size[1] // 32))
    assert rel_err(fast, plain) < 2e-5
    assert ((fast > 0) == (plain > 0)).float().mean().item() >= 0.999


@pytest.mark.parametrize("B,n_words,W", [(2, 6, 96), (8, 20, 96), (16, 20, 128)])
def test_graphed_dmn_replay_equals_eager_forward(B, n_words, W, monkeypatch):
    """One captured HIP graph per shape: replays must reproduce the eager forward bit for bit,
    for new inputs as well (static buffers), and reject other shapes.  (8 queries of 20 words: the
    forward contains the fused mix + layer-0 scan on the matrix cores, captured like any other launch;
    16 queries on a 2 x 4 map with the dynamic filters forced onto the LDS-DMA ring kernel, whose 64 KB
    LDS opt-in must not happen under capture.)"""
    if W == 128:
        monkeypatch.setenv("DMN_DYNFILTER_DMA", "1")
    net = dms_amd.DMN(100, num_filters=10, lang_layers=3, mix_we=True, pretrained=False,
                      high_res=True, upsampling_mode="bilinear", upsampling_amplification=32)
    torch.nn.Module.load_state_dict(net, det_fill_state(net.state_dict()))
    net = net.to(DEV).eval()
    g = dms_amd.GraphedDMN(net, B, 64, W, n_words)
    for seed in (11, 12):
        img = det_uniform((B, 3, 64, W), seed, 1.0).to(DEV)
        words = det_ints((B, n_words), seed + 100, 100).to(DEV)
        with torch.no_grad():
            want = net(img, words)
        got = g(img, words).clone()
        assert got.shape == (B, 1, 64, W) and torch.equal(got, want)
    with pytest.raises(RuntimeError, match="captured for"):
        g(torch.zeros((1, 3, 64, W), device=DEV), torch.zeros((1, n_words), dtype=torch.long, device=DEV))


# ------------------------------------------------------------------------------------
# full-size properties (BASELINE.json configs 2 and 5 shapes; no CPU oracle at this size)
# ------------------------------------------------------------------------------------
def test_head_full_size_winograd_equals_miopen_path_and_shards_by_batch():
    """BASELINE configs[1] head shapes (1005 + [2688, 1552, 704, 336, 64] channels, 16x16 -> 512x512),
    B = 2: Winograd levels vs the plain cat_upsample2x + MIOpen path on the same weights, and
    batch entries independent of their neighbours (what sharding over ranks relies on)."""
    up = dms_amd.UpsamplingModule(1005, mode="bilinear", ker_size=3, amplification=32)
    up.load_state_dict(det_fill_state(up.state_dict()))
    up = up.to(DEV).eval()
    gen = torch.Generator(device=DEV).manual_seed(7)
    x = torch.randn((2, 1005, 16, 16), device=DEV, generator=gen)
    feats = [torch.randn((2, c, s, s), device=DEV, generator=gen)
             for c, s in ((64, 256), (336, 128), (704, 64), (1552, 32), (2688, 16))]
    with torch.no_grad():
        tap = up(x, feats)                                    # default: tap GEMM + gather levels
        tap_one = up(x[1:], [f[1:] for f in feats])
        up.tap_gemm = False
        up.winograd = False
        ref = up(x, feats)
        up.winograd = "auto"
        got = up(x, feats)
        one = up(x[1:], [f[1:] for f in feats])
        up.winograd = "f2"
        got2 = up(x, feats)
    assert got.shape == tap.shape == (2, 1, 512, 512)
    assert rel_err(got, ref) < 5e-5 and rel_err(got2, ref) < 1e-5
    assert rel_err(one, got[1:]) < 1e-5
    assert rel_err(tap, ref) < 1e-5 and rel_err(tap_one, tap[1:]) < 1e-5


def test_sru_full_size_chunked_equals_whole_and_column_subset_oracle():
    """Config-5-like: T=4096, d=1000, k=3 (B reduced to 16 to bound memory/time).
    (i) splitting time in two calls with the carried c equals one call (recurrence
    semantics); (ii) a random subset of columns matches the CPU oracle run on that subset
    only (columns are independent)."""
    T, B, d, k = 4096, 16, 1000, 3
    gen = torch.Generator(device=DEV).manual_seed(1111)
    u = torch.randn((T, B, d, k), device=DEV, generator=gen)
    x = torch.randn((T, B, d), device=DEV, generator=gen)
    bias = 0.1 * torch.randn((2 * d,), device=DEV, generator=gen)
    h, _, cl = ops.sru_forward_raw(u, x, bias, None, "tanh", 1.0, nchunks=1)
    Th = 1500
    h1, _, c1 = ops.sru_forward_raw(u[:Th].contiguous(), x[:Th].contiguous(), bias, None, "tanh", 1.0, nchunks=1)
    h2, _, c2 = ops.sru_forward_raw(u[Th:].contiguous(), x[Th:].contiguous(), bias, c1, "tanh", 1.0, nchunks=1)
    assert torch.equal(torch.cat([h1, h2]), h) and torch.equal(c2, cl)
    cols = torch.tensor([0, 1, 63, 64, 999, 1000, 7777, B * d - 1])
    us = u.view(T, B * d, k)[:, cols].cpu().contiguous().view(T, 1, len(cols), k)
    xs = x.view(T, B * d)[:, cols].cpu().contiguous().view(T, 1, len(cols))
    bj = torch.cat([bias[:d][cols % d], bias[d:][cols % d]]).cpu()
    h_ref, _, _ = sru_ref.sru_fwd_c(us.numpy(), xs.numpy(), bj.numpy(), None, "tanh", 1.0)
    got = h.view(T, B * d)[:, cols].cpu().numpy()
    np.testing.assert_allclose(got, h_ref[:, 0], rtol=1e-4, atol=3e-6)
    # time-parallel variant agrees at full length as well
    h3, _, c3 = ops.sru_forward_raw(u, x, bias, None, "tanh", 1.0, nchunks=16)
    assert rel_err(h3, h) < 1e-5 and rel_err(c3, cl) < 1e-5


def test_recorded_gemm_table_lookup_keeps_results_and_leaves_the_table_alone():
    """ops.use_gemm_table(): TunableOp in look-up mode with the shipped per-shape solution table.
    A shape the table holds (DPN92 conv4 c1x1_a at B=32) must still match an fp64 reference,
    no tuning may happen at run time, and the shipped file must not be rewritten."""
    import hashlib
    import torch.cuda.tunable as tunable
    with open(ops.GEMM_TABLE, "rb") as f:
        before = hashlib.md5(f.read()).hexdigest()
    n = ops.use_gemm_table()
    try:
        assert n > 0, "table rejected: validators do not match this torch/ROCm stack"
        assert tunable.is_enabled() and not tunable.tuning_is_enabled()
        assert os.path.abspath(tunable.get_filename()) != os.path.abspath(ops.GEMM_TABLE)
        gen = torch.Generator(device=DEV).manual_seed(5)
        x = torch.randn((32, 1312, 32, 32), device=DEV, generator=gen)
        w = torch.randn((384, 1312, 1, 1), device=DEV, generator=gen) * 0.05
        got = ops.conv1x1(x, w)
        want = torch.matmul(w.view(384, 1312).double(), x[:2].view(2, 1312, 1024).double()).float()
        assert rel_err(got[:2].reshape(2, 384, 1024), want) < 1e-5
        assert len(tunable.get_results()) == n                # nothing was tuned on the fly
    finally:
        tunable.enable(False)
    with open(ops.GEMM_TABLE, "rb") as f:
        assert hashlib.md5(f.read()).hexdigest() == before


def test_head_reads_backbone_taps_in_place_as_one_concat_operand():
    """DMN links the backbone's pyramid taps to the head: each tap is written behind room for the
    level's own input (DPN.tap_prefix), the level runs ONE GEMM over [x | feat] and the gather
    writes into the next level's buffer.  Must equal the compact-tap / two-GEMM path."""
    net = dms_amd.DMN(100, num_filters=10, lang_layers=3, mix_we=True, pretrained=False,
                      high_res=True, upsampling_mode="bilinear", upsampling_amplification=32)
    torch.nn.Module.load_state_dict(net, det_fill_state(net.state_dict()))
    net = net.to(DEV).eval()
    base = net.langvis.base
    assert base.tap_prefix == {"conv5_3": 1005, "conv4_20": 512, "conv3_4": 256, "conv2_3": 128}
    words = det_ints((2, 6), 152, 100).to(DEV)
    # 128 x 192: every stage takes the in-place path; 128 x 96: the last one falls back (6-wide map),
    # so its tap is compact and that level runs the two-GEMM path next to linked ones
    for size, nlinked in (((128, 192), 4), ((128, 96), 3)):
        img = det_uniform((2, 3) + size, 151, 1.0).to(DEV)
        with torch.no_grad():
            _, feats = base(img)
            for k, (c, p) in enumerate(((336, 128), (704, 256), (1552, 512), (2688, 1005)), 1):
                f = feats[k]
                assert f.shape[1] == c
                if k <= nlinked:
                    assert f._dmn_base.shape[1] == p + c and not f.is_contiguous()
                else:
                    assert f.is_contiguous()
            linked = net(img, words)
            keep, base.tap_prefix = base.tap_prefix, None
            _, feats_c = base(img)
            compact = net(img, words)
            base.tap_prefix = keep
        assert all(a.is_contiguous() for a in feats_c)
        assert all(torch.equal(a, b) for a, b in zip(feats, feats_c))
        assert linked.shape == (2, 1) + size and rel_err(linked, compact) < 1e-5
