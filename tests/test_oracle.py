"""CPU tests (-m "not gpu"): the oracle against the reference's golden vectors and
against itself (C restatement vs torch restatement, hand-written backward vs autograd)."""
import numpy as np
import pytest
import torch

from oracle import dmn_ref, sru_ref
from oracle.detgen import det_fill_state, det_uniform
from tests.util import base_inputs, cfg_from_golden, load_golden, rel_err


def _rand_case(T, B, d, k, seed):
    u = det_uniform((T, B, d, k), seed, 1.5)
    x = det_uniform((T, B, d), seed + 1, 1.0) if k == 3 else None
    bias = det_uniform((2 * d,), seed + 2, 0.5)
    c0 = det_uniform((B, d), seed + 3, 1.0)
    return u, x, bias, c0


@pytest.mark.parametrize("T,B,d,k", [(1, 1, 5, 3), (7, 3, 5, 4), (64, 1, 64, 3), (33, 2, 21, 4),
                                     (0, 2, 4, 3)])
@pytest.mark.parametrize("act", ["tanh", "relu", "identity"])
def test_c_oracle_matches_torch_restatement_fwd(T, B, d, k, act):
    u, x, bias, c0 = _rand_case(T, B, d, k, 11)
    h_t, c_t = sru_ref.sru_scan_torch(u, x, bias, c0, act, 1.0)
    h_c, c_c, cl = sru_ref.sru_fwd_c(u.numpy(), None if x is None else x.numpy(), bias.numpy(),
                                     c0.numpy(), act, 1.0)
    assert h_c.shape == (T, B, d)
    if T:
        np.testing.assert_allclose(h_c, h_t.numpy(), rtol=2e-6, atol=2e-6)
        np.testing.assert_allclose(c_c, c_t.numpy(), rtol=2e-6, atol=2e-6)
        np.testing.assert_allclose(cl, c_t[-1].numpy(), rtol=2e-6, atol=2e-6)
    else:
        np.testing.assert_array_equal(cl, c0.numpy())


@pytest.mark.parametrize("T,B,d,k", [(1, 1, 5, 3), (9, 3, 5, 4), (40, 2, 17, 3)])
@pytest.mark.parametrize("act,scale_x", [("tanh", 1.0), ("relu", 1.0), ("tanh", 1.7)])
def test_c_oracle_bwd_matches_autograd(T, B, d, k, act, scale_x):
    u, x, bias, c0 = _rand_case(T, B, d, k, 23)
    gh = det_uniform((T, B, d), 31, 1.0)
    gl = det_uniform((B, d), 32, 1.0)
    leaves = [t.clone().double().requires_grad_(True) for t in (u, bias, c0)]
    xd = x.clone().double().requires_grad_(True) if x is not None else None
    h, c = sru_ref.sru_scan_torch(leaves[0], xd, leaves[1], leaves[2], act, scale_x)
    loss = (h * gh.double()).sum() + (c[-1] * gl.double()).sum()
    loss.backward()
    _, c_all, _ = sru_ref.sru_fwd_c(u.numpy(), None if x is None else x.numpy(), bias.numpy(),
                                    c0.numpy(), act, scale_x)
    gu, gx, gb, gc0 = sru_ref.sru_bwd_c(u.numpy(), None if x is None else x.numpy(),
                                        bias.numpy(), c0.numpy(), c_all, gh.numpy(), gl.numpy(),
                                        act, scale_x)
    np.testing.assert_allclose(gu, leaves[0].grad.float().numpy(), rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(gb.sum(1).reshape(-1), leaves[1].grad.float().numpy(), rtol=2e-4, atol=5e-5)
    np.testing.assert_allclose(gc0, leaves[2].grad.float().numpy(), rtol=2e-4, atol=2e-5)
    if x is not None:
        np.testing.assert_allclose(gx, xd.grad.float().numpy(), rtol=2e-4, atol=2e-5)


def test_spatial_matches_reference_golden():
    g = load_golden("spatial")
    for key in g.files:
        _, H, W = key.split("_")
        np.testing.assert_array_equal(dmn_ref.spatial_batch(int(H), int(W)).numpy(), g[key])


def _oracle_base(name, use_c=False):
    from dms_amd import BaseDMN          # only for parameter names/shapes (CPU construction)
    g = load_golden(name)
    cfg = cfg_from_golden(g)
    H, W, L, D = int(g["H"]), int(g["W"]), int(g["L"]), int(g["dict_size"])
    keys = [k for k in g["keys"]]
    shapes = {k: eval(s) for k, s in zip(g["keys"], g["shapes"])}
    state = det_fill_state({k: torch.empty(shapes[k]) for k in keys})
    P = {"langvis." + k: v for k, v in state.items()}
    vis, lang = base_inputs(cfg, H, W, L, D)
    with torch.no_grad():
        out = dmn_ref.hot_path_single(P, vis, lang, cfg, use_c=use_c)
    return out, g


@pytest.mark.parametrize("name", ["base_tiny_k3", "base_tiny_k4", "base_tiny_k4_nowe_hr"])
def test_oracle_hot_path_matches_reference_tiny(name):
    out, g = _oracle_base(name)
    assert out.shape == g["out"].shape
    assert rel_err(out, g["out"]) < 1e-5


def test_oracle_hot_path_matches_reference_full_320(tmp_path):
    # config 1's post-backbone shapes (10x10, L=20, full channel counts); C oracle scan
    out, g = _oracle_base("base_full_320_lowres", use_c=True)
    assert tuple(out.shape) == (1, 1, 10, 10)
    assert rel_err(out, g["out"]) < 2e-5


def test_oracle_upsampling_matches_reference():
    g = load_golden("upsample_small")
    shapes = {k: eval(s) for k, s in zip(g["keys"], g["shapes"])}
    st = det_fill_state({k: torch.empty(shapes[k]) for k in g["keys"]})
    P = {"upsample." + k: v for k, v in st.items()}
    fch = [6, 5, 4, 3, 2]
    x = det_uniform((1, 7, 3, 4), 303, 1.0)
    feats = [None, None, det_uniform((1, fch[2], 12, 16), 306, 1.0),
             det_uniform((1, fch[1], 6, 8), 305, 1.0), det_uniform((1, fch[0], 3, 4), 304, 1.0)]
    with torch.no_grad():
        y = dmn_ref.upsampling(P, x, feats, 3)
    assert rel_err(y, g["out"]) < 1e-5
    feats = [None, None, det_uniform((1, fch[2], 13, 15), 309, 1.0),
             det_uniform((1, fch[1], 7, 9), 308, 1.0), det_uniform((1, fch[0], 4, 5), 307, 1.0)]
    with torch.no_grad():
        y = dmn_ref.upsampling(P, x, feats, 3)
    assert rel_err(y, g["out_mis"]) < 1e-5
