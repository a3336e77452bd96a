"""GPU parity tests (-m gpu), part 2: BASELINE.json's configurations at their REAL sizes, the
training-step gradients, shape fuzz and shard invariance.

  configs[1]  DMN high-res 512 px, batch 32: whole forward vs the CPU oracle on samples != 0
  configs[2]  batch sharding: every HIP kernel is bitwise invariant to how the batch is split
  configs[3]  training step (BaseDMN.hot_path -> bilinear upsample -> BCEWithLogits,
              train.py:275-283): every parameter's gradient vs autograd of the torch oracle
  configs[4]  standalone scan T=4096, B=128, d=1000 (the real batch), column-subset oracle
  SURVEY 4(v) hypothesis-driven shape fuzz of dmn_sru_fwd
  SURVEY 8f4  left-padded mixed-length queries through the HIP language SRU == batch-1 runs
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F
from hypothesis import HealthCheck, given, settings, strategies as st

import dms_amd
from dms_amd import _lib, ops
from oracle import dmn_ref, sru_ref
from oracle.detgen import det_fill_state, det_ints, det_uniform
from tests.util import assert_close_elementwise, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), "these tests need a real MI355X"
    _lib.load()


# ------------------------------------------------------------------------------------
# configs[1]: the benchmarked shape itself
# ------------------------------------------------------------------------------------
def test_config2_dmn_highres_512px_batch32_matches_cpu_oracle_on_samples_inside_the_batch():
    """DMN high-res 512 px at B = 32 (what bench.py times): samples 7 and 29 of the batch against
    torch-CPU DPN92 + oracle/dmn_ref (reference-shaped hot path, plain-C scan) + oracle upsampling.
    Logits within 1e-4 relative (norm-wise AND element-wise), thresholded masks >= 99.9 %."""
    torch.manual_seed(1111)                                       # reference default seed (train.py:75)
    net = dms_amd.DMN(1000, num_filters=10, lang_layers=3, mix_we=True, pretrained=False,
                      high_res=True, upsampling_mode="bilinear", upsampling_amplification=32).eval()
    P = {k: v.clone() for k, v in net.state_dict().items()}
    cpu_base = net.langvis.base                                   # torch path on CPU tensors
    g = torch.Generator().manual_seed(1112)
    B, picks = 32, (7, 29)
    img = torch.randn((B, 3, 512, 512), generator=g)
    words = torch.randint(0, 1000, (B, 20), generator=g)
    cfg = dict(num_filters=10, vis_size=2688, lang_layers=3, mixed_layers=3, mix_we=True, high_res=True)
    want = {}
    sru_ref.set_threads(min(os.cpu_count() or 1, 16))
    with torch.no_grad():
        for b in picks:
            vis, feats = cpu_base(img[b:b + 1])
            x = dmn_ref.hot_path_single(P, vis, words[b:b + 1], cfg, use_c=True)
            want[b] = dmn_ref.upsampling(P, x, feats, 5)
    net = net.to(DEV)
    with torch.no_grad():
        out = net(img.to(DEV), words.to(DEV)).cpu()
    assert out.shape == (B, 1, 512, 512) and bool(torch.isfinite(out).all())
    for b in picks:
        assert rel_err(out[b:b + 1], want[b]) < 1e-4, b
        assert_close_elementwise(out[b:b + 1], want[b])
        agree = ((out[b:b + 1] > 0) == (want[b] > 0)).float().mean().item()
        assert agree >= 0.999, (b, agree)
    assert not torch.equal(out[7], out[29])


# ------------------------------------------------------------------------------------
# configs[4]: the stress scan at its real batch
# ------------------------------------------------------------------------------------
def test_config5_scan_T4096_B128_d1000_split_in_time_and_column_subset_oracle():
    """T = 64*64, B = 128, d = 1000, k = 3 (128 000 columns = 2000 wavefronts: the depth-8 ring the
    small-shape tests only reach through DMN_SRU_DEPTH).  (i) two calls split in time with the
    carried c equal one call bit for bit; (ii) columns spread over the whole batch match the CPU
    oracle run on those columns alone (columns are independent)."""
    T, B, d, k = 4096, 128, 1000, 3
    gen = torch.Generator(device=DEV).manual_seed(1111)
    u = torch.randn((T, B, d, k), device=DEV, generator=gen)
    x = torch.randn((T, B, d), device=DEV, generator=gen)
    bias = 0.1 * torch.randn((2 * d,), device=DEV, generator=gen)
    h, _, cl = ops.sru_forward_raw(u, x, bias, None, "tanh", 1.0, nchunks=1)
    Th = 1500
    h1, _, c1 = ops.sru_forward_raw(u[:Th], x[:Th], bias, None, "tanh", 1.0, nchunks=1)
    h2, _, c2 = ops.sru_forward_raw(u[Th:], x[Th:], bias, c1, "tanh", 1.0, nchunks=1)
    assert torch.equal(h1, h[:Th]) and torch.equal(h2, h[Th:]) and torch.equal(c2, cl)
    del h1, h2
    cols = torch.tensor([0, 1, 63, 64, 999, 1000, 7777, 63999, 64000, 100001, B * d - 65, B * d - 1])
    us = u.view(T, B * d, k)[:, cols.to(DEV)].cpu().contiguous().view(T, 1, len(cols), k)
    xs = x.view(T, B * d)[:, cols.to(DEV)].cpu().contiguous().view(T, 1, len(cols))
    bj = torch.cat([bias[:d].cpu()[cols % d], bias[d:].cpu()[cols % d]])
    h_ref, _, cl_ref = sru_ref.sru_fwd_c(us.numpy(), xs.numpy(), bj.numpy(), None, "tanh", 1.0)
    np.testing.assert_allclose(h.view(T, B * d)[:, cols.to(DEV)].cpu().numpy(), h_ref[:, 0], rtol=1e-4, atol=3e-6)
    np.testing.assert_allclose(cl.view(B * d)[cols.to(DEV)].cpu().numpy(), cl_ref[0], rtol=1e-4, atol=3e-6)


# ------------------------------------------------------------------------------------
# configs[3]: gradients of the training step
# ------------------------------------------------------------------------------------
@pytest.mark.parametrize("hid_mixed,mix_we,fold", [(16, True, True), (21, True, True), (21, False, True), (16, False, True),
                                                   (21, True, False)])
def test_training_step_gradients_match_autograd_of_the_torch_oracle(hid_mixed, mix_we, fold):
    """BaseDMN.hot_path -> bilinear upsample to the mask size -> BCEWithLogitsLoss (train.py:275-283)
    through the HIP forward + hand-written backward kernels (scan k = 3 and k = 4, dynamic filters,
    concat-free mix -- with comb_conv folded into layer 0's projection for k = 4, and without) vs torch autograd of oracle/dmn_ref.hot_path (pure torch, reference-shaped).
    The loss and EVERY parameter's gradient (and the gradient w.r.t. the backbone features) within
    1e-4 relative."""
    cfg = dict(emb_size=16, hid_size=16, vis_size=32, num_filters=3, mixed_size=16,
               hid_mixed_size=hid_mixed, lang_layers=2, mixed_layers=3, mix_we=mix_we, high_res=False)
    net = dms_amd.BaseDMN(50, pretrained=False, **cfg)
    net.base = torch.nn.Identity()
    state = det_fill_state(net.state_dict())
    torch.nn.Module.load_state_dict(net, state)
    net = net.to(DEV).train()
    net.fold_in_training = fold       # k = 4 layer 0 (hid_mixed != mixed_size): comb_conv folded into its projection
    B, H, W, L = 2, 4, 3, 5
    vis = det_uniform((B, cfg["vis_size"], H, W), 601, 1.0).clamp_(min=0)
    lang = det_ints((B, L), 602, 50)
    mask = (det_uniform((B, 1, 8 * H, 8 * W), 603, 1.0) > 0).float()
    crit = torch.nn.BCEWithLogitsLoss()

    vg = vis.to(DEV).requires_grad_(True)
    out = net.hot_path(vg, lang.to(DEV))
    loss = crit(F.interpolate(out, size=mask.shape[-2:], mode="bilinear", align_corners=False), mask.to(DEV))
    loss.backward()

    P = {"langvis." + k: v.clone().requires_grad_(True) for k, v in state.items()}
    vr = vis.clone().requires_grad_(True)
    ref = dmn_ref.hot_path(P, vr, lang, cfg, use_c=False)
    loss_r = crit(F.interpolate(ref, size=mask.shape[-2:], mode="bilinear", align_corners=False), mask)
    loss_r.backward()

    assert rel_err(out.detach().cpu(), ref.detach()) < 1e-5
    assert abs(float(loss) - float(loss_r)) < 1e-5 * abs(float(loss_r))
    assert rel_err(vg.grad.cpu(), vr.grad) < 1e-4
    checked = 0
    for name, p in net.named_parameters():
        assert p.grad is not None, name
        gr = P["langvis." + name].grad
        assert gr is not None and float(gr.abs().max()) > 0, name
        assert rel_err(p.grad.cpu(), gr) < 1e-4, name
        checked += 1
    assert checked == len(state)


def test_one_training_step_of_the_whole_dmn_is_finite_and_moves_the_parameters():
    """One step of configs[3]'s loop at small size (low-res DMN incl. DPN92, 96 x 128 px, B = 2, train()
    mode, Adam lr 1e-5): loss finite, every parameter receives a finite gradient, parameters move;
    and the hot path's parameter gradients equal those of the same step taken through the PLAIN path
    (backbone features detached and fed to the torch oracle)."""
    torch.manual_seed(7)
    net = dms_amd.DMN(60, num_filters=10, lang_layers=3, mix_we=True, pretrained=False, high_res=False)
    net = net.to(DEV).train()
    opt = torch.optim.Adam(net.parameters(), lr=1e-5)
    img = det_uniform((2, 3, 96, 128), 611, 1.0).to(DEV)
    words = det_ints((2, 6), 612, 60).to(DEV)
    mask = (det_uniform((2, 1, 96, 128), 613, 1.0) > 0).float().to(DEV)
    crit = torch.nn.BCEWithLogitsLoss()
    before = {k: v.detach().clone() for k, v in net.named_parameters()}
    opt.zero_grad(set_to_none=True)
    out = net(img, words)
    assert out.shape == (2, 1, 3, 4)
    loss = crit(F.interpolate(out, size=mask.shape[-2:], mode="bilinear", align_corners=False), mask)
    loss.backward()
    assert bool(torch.isfinite(loss))
    grads = {k: p.grad.detach().clone() for k, p in net.named_parameters()}
    assert all(bool(torch.isfinite(g).all()) for g in grads.values())
    assert float(grads["langvis.mrnn.rnn_lst.0.weight"].abs().max()) > 0
    assert float(grads["langvis.base.features.conv1_1.conv.weight"].abs().max()) > 0
    opt.step()
    moved = sum(int(not torch.equal(p.detach(), before[k])) for k, p in net.named_parameters())
    assert moved >= 0.9 * len(before)
    # plain path: same (pre-step) weights, backbone features from the product, hot path = torch oracle on CPU
    torch.nn.Module.load_state_dict(net, {**net.state_dict(), **before})
    with torch.no_grad():
        vis, _ = net.langvis.base(img)
    P = {k: v.detach().cpu().clone().requires_grad_(k.startswith("langvis.") and ".base." not in k)
         for k, v in net.state_dict().items()}
    cfg = dict(num_filters=10, vis_size=2688, lang_layers=3, mixed_layers=3, mix_we=True, high_res=False)
    ref = dmn_ref.hot_path(P, vis.cpu(), words.cpu(), cfg, use_c=False)
    crit(F.interpolate(ref, size=mask.shape[-2:], mode="bilinear", align_corners=False), mask.cpu()).backward()
    for k in ("langvis.emb.weight", "langvis.adaptative_filter.weight", "langvis.comb_conv.weight",
              "langvis.lang_model.rnn_lst.2.weight", "langvis.mrnn.rnn_lst.0.weight", "langvis.mrnn.rnn_lst.0.bias",
              "langvis.mrnn.rnn_lst.2.weight", "langvis.mrnn.rnn_lst.2.bias", "langvis.output_collapse.weight"):
        assert rel_err(grads[k].cpu(), P[k].grad) < 1e-4, k


# ------------------------------------------------------------------------------------
# shape fuzz
# ------------------------------------------------------------------------------------
@settings(max_examples=60, deadline=None, derandomize=True,
          suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(T=st.integers(0, 75), B=st.integers(1, 4), d=st.integers(1, 150), k=st.sampled_from([3, 4]),
       act=st.sampled_from(["tanh", "relu", "identity"]), scale_x=st.sampled_from([1.0, 0.5, 1.7]),
       depth=st.sampled_from(["0", "3", "4", "6", "8", "16", "24", "32"]), h_frac=st.sampled_from([0.0, 0.34, 1.0]),
       with_c0=st.booleans(), want_c=st.booleans(), seed=st.integers(0, 10 ** 6))
def test_sru_fwd_shape_fuzz_matches_oracle(T, B, d, k, act, scale_x, depth, h_frac, with_c0, want_c, seed):
    """Odd d, T shorter than the register ring, B*d not a multiple of 64, every ring depth, tails."""
    os.environ["DMN_SRU_DEPTH"] = depth
    try:
        u = det_uniform((T, B, d, k), seed, 1.5)
        x = det_uniform((T, B, d), seed + 1, 1.0) if k == 3 else None
        bias = det_uniform((2 * d,), seed + 2, 0.5)
        c0 = det_uniform((B, d), seed + 3, 1.0) if with_c0 else None
        h_t0 = 0 if want_c else int(h_frac * T)
        h_ref, c_ref, cl_ref = sru_ref.sru_fwd_c(u.numpy(), None if x is None else x.numpy(), bias.numpy(),
                                                 None if c0 is None else c0.numpy(), act, scale_x)
        h, c_all, c_last = ops.sru_forward_raw(u.to(DEV), None if x is None else x.to(DEV), bias.to(DEV),
                                               None if c0 is None else c0.to(DEV), act, scale_x, h_t0,
                                               want_c=want_c, nchunks=1)
        assert h.shape == (T - h_t0, B, d)
        np.testing.assert_allclose(h.cpu().numpy(), h_ref[h_t0:], rtol=1e-4, atol=3e-6)
        np.testing.assert_allclose(c_last.cpu().numpy(), cl_ref, rtol=1e-4, atol=3e-6)
        if want_c:
            np.testing.assert_allclose(c_all.cpu().numpy(), c_ref, rtol=1e-4, atol=3e-6)
    finally:
        os.environ.pop("DMN_SRU_DEPTH", None)


@settings(max_examples=25, deadline=None, derandomize=True,
          suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(T=st.integers(0, 60), B=st.integers(1, 3), d=st.integers(1, 140), k=st.sampled_from([3, 4]),
       act=st.sampled_from(["tanh", "relu", "identity"]), scale_x=st.sampled_from([1.0, 1.7]),
       seed=st.integers(0, 10 ** 6))
def test_sru_bwd_shape_fuzz_matches_oracle(T, B, d, k, act, scale_x, seed):
    u = det_uniform((T, B, d, k), seed, 1.5)
    x = det_uniform((T, B, d), seed + 1, 1.0) if k == 3 else None
    bias = det_uniform((2 * d,), seed + 2, 0.5)
    c0 = det_uniform((B, d), seed + 3, 1.0)
    gh, gl = det_uniform((T, B, d), seed + 4, 1.0), det_uniform((B, d), seed + 5, 1.0)
    xn = None if x is None else x.numpy()
    _, c_all, _ = sru_ref.sru_fwd_c(u.numpy(), xn, bias.numpy(), c0.numpy(), act, scale_x)
    gu_r, gx_r, gb_r, gc0_r = sru_ref.sru_bwd_c(u.numpy(), xn, bias.numpy(), c0.numpy(), c_all, gh.numpy(),
                                                gl.numpy(), act, scale_x)
    ug, bg, cg = (t.to(DEV).requires_grad_(True) for t in (u, bias, c0))
    xg = None if x is None else x.to(DEV).requires_grad_(True)
    h, c_last = ops.sru_compute(ug, xg, bg, cg, act, scale_x)
    ((h * gh.to(DEV)).sum() + (c_last * gl.to(DEV)).sum()).backward()
    np.testing.assert_allclose(ug.grad.cpu().numpy(), gu_r, rtol=2e-4, atol=1e-5)
    np.testing.assert_allclose(bg.grad.cpu().numpy(), gb_r.sum(1).reshape(-1), rtol=2e-4, atol=5e-5)
    np.testing.assert_allclose(cg.grad.cpu().numpy(), gc0_r, rtol=2e-4, atol=1e-5)
    if x is not None:
        np.testing.assert_allclose(xg.grad.cpu().numpy(), gx_r, rtol=2e-4, atol=1e-5)


@settings(max_examples=30, deadline=None, derandomize=True,
          suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(B=st.integers(1, 5), S=st.integers(1, 41), d=st.integers(1, 200), model_shape=st.booleans(),
       act=st.sampled_from(["tanh", "relu", "identity"]), with_c0=st.booleans(), seed=st.integers(0, 10 ** 6))
def test_mix_sru_fwd_shape_fuzz_is_bit_identical_to_mix_then_scan(B, S, d, model_shape, act, with_c0, seed):
    """dmn_mix_sru_fwd over odd shapes: the model's K = 10 / L = 20 (the two-wave kernels: VALU and matrix-core
    variant, partial last super-blocks, ragged and shifted column windows) and other K / L (single-wave kernel).
    Every variant must reproduce dmn_mix_bcast_add + dmn_sru_fwd bit for bit."""
    L, K = (20, 10) if model_shape else (1 + seed % 7, 1 + seed % 12)
    M4 = 4 * d
    A = det_uniform((B, S, M4), seed, 1.0).to(DEV)
    Bt = det_uniform((L, B, M4), seed + 1, 1.0).to(DEV)
    P = det_uniform((L, B, K, S), seed + 2, 1.0).to(DEV)
    Wp = det_uniform((M4, K), seed + 3, 0.5).to(DEV)
    bias = det_uniform((2 * d,), seed + 4, 0.5).to(DEV)
    c0 = det_uniform((B, d), seed + 5, 1.0).to(DEV) if with_c0 else None
    U0 = ops.mix_bcast_add(A, Bt, P, Wp).view(S * L, B, d, 4)
    h_ref, _, c_ref = ops.sru_forward_raw(U0, None, bias, c0, act, 1.0, nchunks=1)
    for force in (None, "0", "1"):
        if force is not None:
            os.environ["DMN_MIX_SRU_MFMA"] = force
        try:
            h, c = ops.mix_sru_fwd(A, Bt, P, Wp, bias, c0, act)
        finally:
            os.environ.pop("DMN_MIX_SRU_MFMA", None)
        assert torch.equal(h, h_ref) and torch.equal(c, c_ref), "DMN_MIX_SRU_MFMA=%s" % force


@pytest.mark.parametrize("use_tanh,use_relu,scale_x", [(0, 0, 1.0), (0, 1, 1.7), (1, 0, 0.5)])
def test_sru_module_with_other_activation_and_highway_scale_matches_oracle(use_tanh, use_relu, scale_x):
    """The upstream unknowns as MODULE options (use_tanh / use_relu / scale_x), fwd + autograd."""
    act = "relu" if use_relu else ("tanh" if use_tanh else "identity")
    rnn = dms_amd.SRU(10, 10, num_layers=2, use_tanh=use_tanh, use_relu=use_relu, scale_x=scale_x)
    rnn.load_state_dict(det_fill_state(rnn.state_dict()))
    ref = sru_ref.SRURef(10, 10, num_layers=2, activation=act, scale_x=scale_x)
    ref.load_state_dict(rnn.state_dict())
    rnn = rnn.to(DEV)
    x = det_uniform((13, 3, 10), 621, 1.0)
    xg, xr = x.to(DEV).requires_grad_(True), x.clone().requires_grad_(True)
    h, c = rnn(xg)
    hr, cr = ref(xr)
    assert rel_err(h.detach().cpu(), hr.detach()) < 1e-5 and rel_err(c.detach().cpu(), cr.detach()) < 1e-5
    w = det_uniform((13, 3, 10), 622, 1.0)
    (h * w.to(DEV)).sum().backward()
    (hr * w).sum().backward()
    assert rel_err(xg.grad.cpu(), xr.grad) < 1e-4
    for (n, p), (_, pr) in zip(rnn.named_parameters(), ref.named_parameters()):
        assert rel_err(p.grad.cpu(), pr.grad) < 1e-4, n
    with torch.no_grad():                                       # inference kernels (no c store, tail)
        h_inf, _ = rnn(xg, tail=4)
    assert rel_err(h_inf.cpu(), hr.detach()[-4:]) < 1e-5


# ------------------------------------------------------------------------------------
# configs[2]: sharding the batch over ranks changes no bit of any HIP kernel's output
# ------------------------------------------------------------------------------------
def test_hip_kernels_are_bitwise_invariant_to_how_the_batch_is_sharded():
    """an N-way shard must reproduce the 1-way result bitwise, per sample.  Checked for
    every hand-written kernel of the inference path at full channel counts on B = 4 vs 2 + 2 vs
    1 + 1 + 1 + 1 (the library GEMMs in between pick kernels per shape and are covered to 1e-5 by the
    model-level tests)."""
    B, C8, S, L, K, d = 4, 2696, 64, 20, 10, 1005
    V = det_uniform((B, C8, S), 701, 1.0).to(DEV)
    Fi = torch.sigmoid(det_uniform((L, B, K, C8), 702, 2.0)).to(DEV)
    A = det_uniform((B, S, 4 * d), 703, 1.0).to(DEV)
    Bt = det_uniform((L, B, 4 * d), 704, 1.0).to(DEV)
    Wp = det_uniform((4 * d, K), 705, 0.3).to(DEV)
    bias = det_uniform((2 * d,), 706, 0.5).to(DEV)
    T = S * L
    x1 = det_uniform((T, B, d), 707, 1.0).to(DEV)
    w1 = det_uniform((d, 3 * d), 708, 0.05).to(DEV)
    Z = det_uniform((B, 9, 16, 16, 16), 709, 1.0).to(DEV)
    zb = det_uniform((16,), 710, 0.5).to(DEV)
    vis = det_uniform((B, 2688, 4, 4), 711, 1.0).to(DEV)

    def run(sl):
        P = ops.dynfilter(V[sl].contiguous(), Fi[:, sl].contiguous())
        h0, c0 = ops.mix_sru_fwd(A[sl].contiguous(), Bt[:, sl].contiguous(), P, Wp, bias, None, "tanh")
        U0 = ops.mix_bcast_add(A[sl].contiguous(), Bt[:, sl].contiguous(), P, Wp)
        xs = x1[:, sl].contiguous()
        u1 = torch.stack([xs[:, b].mm(w1) for b in range(xs.shape[1])], 1).view(T, -1, d, 3)   # per-sample GEMM
        h1, _, c1 = ops.sru_forward_raw(u1, xs, bias, None, "tanh", 1.0, T - S, nchunks=1)
        cs = ops.sru_state(u1[..., :2].contiguous(), bias, None, nchunks=1)
        up = ops.tap_upsample_gather(Z[sl].contiguous(), zb)
        cc = ops.cat_coords(vis[sl].contiguous())
        return dict(P=P.transpose(0, 1), h0=h0.transpose(0, 1), c0=c0, U0=U0.transpose(0, 1),
                    h1=h1.transpose(0, 1), c1=c1, cs=cs, up=up, cc=cc)

    whole = run(slice(0, 4))
    for parts in ([slice(0, 2), slice(2, 4)], [slice(i, i + 1) for i in range(4)]):
        shards = [run(sl) for sl in parts]
        for name, full in whole.items():
            got = torch.cat([s[name] for s in shards], 0)
            assert torch.equal(got, full), (name, len(parts))


# ------------------------------------------------------------------------------------
# variable-length queries
# ------------------------------------------------------------------------------------
def test_left_padded_query_batch_through_hip_language_sru_equals_batch1_runs_bit_for_bit():
    """Corpus.tokenize_batch left-pads mixed-length queries with <pad> (word_utils.py:76-80) and the
    language SRU runs over the pads like over any word (dmn.py:147-156): the batched HIP encoder must
    give, per query, exactly the bits of that query's own batch-1 run (the reference's only mode)."""
    from dms_amd.data import QuerySet
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "refdata")
    qs = QuerySet(root, "unc", "trainval")
    ids = qs.query_batch(range(len(qs)), device=DEV)                      # (11, 20), 1 .. 20+ real tokens
    pad = qs.corpus.dictionary["<pad>"]
    n_real = (ids != pad).sum(1).tolist()
    assert ids.shape == (len(qs), 20) and min(n_real) <= 2 and max(n_real) == 20
    emb = torch.nn.Embedding(len(qs.corpus), 96)
    rnn = dms_amd.SRU(96, 96, num_layers=3)
    emb.load_state_dict(det_fill_state(emb.state_dict()))
    rnn.load_state_dict(det_fill_state(rnn.state_dict()))
    emb, rnn = emb.to(DEV), rnn.to(DEV).eval()
    with torch.no_grad():
        we = emb(ids).transpose(0, 1).contiguous()                        # (L, B, E)  dmn.py:147-149
        h, c = rnn(we)
        for b in range(ids.shape[0]):
            h1, c1 = rnn(we[:, b:b + 1].contiguous())
            # the x.W projection is a per-row dot product whichever kernel the library picks for the
            # shape; compare the scan bit for bit on the SAME projected input as well
            assert rel_err(h[:, b:b + 1], h1) < 1e-6 and rel_err(c[:, b:b + 1], c1) < 1e-6
        # scan kernel alone, identical U: bitwise
        cell = rnn.rnn_lst[0]
        u = we.view(-1, 96).mm(cell.weight).view(20, -1, 96, 3)
        hb, _, cb = ops.sru_forward_raw(u, we, cell.bias, None, "tanh", 1.0, nchunks=1)
        for b in range(ids.shape[0]):
            h1, _, c1 = ops.sru_forward_raw(u[:, b:b + 1].contiguous(), we[:, b:b + 1].contiguous(),
                                            cell.bias, None, "tanh", 1.0, nchunks=1)
            assert torch.equal(hb[:, b:b + 1], h1) and torch.equal(cb[b:b + 1], c1)
    # and the oracle agrees on a padded query
    ref = sru_ref.SRURef(96, 96, num_layers=3)
    ref.load_state_dict({k: v.cpu() for k, v in rnn.state_dict().items()})
    with torch.no_grad():
        hr, _ = ref(we[:, 3:4].cpu())
    assert rel_err(h[:, 3:4].cpu(), hr) < 1e-5


# ------------------------------------------------------------------------------------
# configs[3]: the training step's scans at their real size
# ------------------------------------------------------------------------------------
@pytest.mark.parametrize("k", [3, 4])
def test_config4_training_scans_T2000_B64_d1005_forward_with_c_and_backward_column_subset_oracle(k):
    """mSRU of the 320 px training step: T = 100 pixels x 20 words, B = 64, d = 1005 (64 320 columns =
    1005 wavefronts, one per SIMD: the occupancy and memory latencies the small-shape tests never
    see).  Forward with the per-step c store, then the backward scan, on columns spread over the
    batch vs the CPU oracle run on those columns alone; grad_bias vs the sum of the oracle's
    per-column terms over a whole j-class."""
    T, B, d = 2000, 64, 1005
    gen = torch.Generator(device=DEV).manual_seed(2222 + k)
    u = torch.randn((T, B, d, k), device=DEV, generator=gen)
    x = torch.randn((T, B, d), device=DEV, generator=gen) if k == 3 else None
    bias = 0.1 * torch.randn((2 * d,), device=DEV, generator=gen)
    c0 = torch.randn((B, d), device=DEV, generator=gen)
    gh = torch.randn((T, B, d), device=DEV, generator=gen)
    gl = torch.randn((B, d), device=DEV, generator=gen)
    ug, bg, cg = u.clone().requires_grad_(True), bias.clone().requires_grad_(True), c0.clone().requires_grad_(True)
    xg = x.clone().requires_grad_(True) if x is not None else None
    h, c_last = ops.sru_compute(ug, xg, bg, cg, "tanh", 1.0)
    ((h * gh).sum() + (c_last * gl).sum()).backward()
    assert all(bool(torch.isfinite(t.grad).all()) for t in (ug, bg, cg))
    # columns: first / last of wavefronts, sample boundaries, and ALL 64 samples of column j = 17
    jcls = torch.arange(B) * d + 17
    cols = torch.cat([torch.tensor([0, 63, 64, 1004, 1005, 32159, 32160, B * d - 64, B * d - 1]), jcls])
    cd = cols.to(DEV)
    n = len(cols)
    us = u.view(T, B * d, k)[:, cd].cpu().contiguous().view(T, 1, n, k)
    xs = x.view(T, B * d)[:, cd].cpu().contiguous().view(T, 1, n) if x is not None else None
    bj = torch.cat([bias[:d].cpu()[cols % d], bias[d:].cpu()[cols % d]])
    c0s = c0.view(B * d)[cd].cpu().view(1, n)
    ghs = gh.view(T, B * d)[:, cd].cpu().contiguous().view(T, 1, n)
    gls = gl.view(B * d)[cd].cpu().view(1, n)
    xn = None if xs is None else xs.numpy()
    h_ref, c_ref, cl_ref = sru_ref.sru_fwd_c(us.numpy(), xn, bj.numpy(), c0s.numpy(), "tanh", 1.0)
    np.testing.assert_allclose(h.detach().view(T, B * d)[:, cd].cpu().numpy(), h_ref[:, 0], rtol=1e-4, atol=3e-6)
    np.testing.assert_allclose(c_last.detach().view(B * d)[cd].cpu().numpy(), cl_ref[0], rtol=1e-4, atol=3e-6)
    gu_r, gx_r, gb_r, gc0_r = sru_ref.sru_bwd_c(us.numpy(), xn, bj.numpy(), c0s.numpy(), c_ref, ghs.numpy(),
                                                gls.numpy(), "tanh", 1.0)
    np.testing.assert_allclose(ug.grad.view(T, B * d, k)[:, cd].cpu().numpy(), gu_r[:, 0], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(cg.grad.view(B * d)[cd].cpu().numpy(), gc0_r[0], rtol=2e-4, atol=2e-5)
    if x is not None:
        np.testing.assert_allclose(xg.grad.view(T, B * d)[:, cd].cpu().numpy(), gx_r[:, 0], rtol=2e-4, atol=2e-5)
    # in-kernel batch reduction: grad_bias[17] and grad_bias[d + 17] = sums over the 64 samples
    want_b = gb_r[:, 0, 9:].sum(axis=1)
    got_b = torch.stack([bg.grad[17], bg.grad[d + 17]]).cpu().numpy()
    np.testing.assert_allclose(got_b, want_b, rtol=5e-4, atol=1e-3)
    # the batch reduction runs in sample order: a second backward on the same inputs gives the same bits
    ug2, bg2, cg2 = u.clone().requires_grad_(True), bias.clone().requires_grad_(True), c0.clone().requires_grad_(True)
    xg2 = x.clone().requires_grad_(True) if x is not None else None
    h2, cl2 = ops.sru_compute(ug2, xg2, bg2, cg2, "tanh", 1.0)
    ((h2 * gh).sum() + (cl2 * gl).sum()).backward()
    assert torch.equal(bg2.grad, bg.grad) and torch.equal(ug2.grad, ug.grad)
    # ... and the atomic form (DMN_SRU_BWD_ATOMIC=1) agrees with it to rounding
    os.environ["DMN_SRU_BWD_ATOMIC"] = "1"
    try:
        ug3, bg3, cg3 = u.clone().requires_grad_(True), bias.clone().requires_grad_(True), c0.clone().requires_grad_(True)
        xg3 = x.clone().requires_grad_(True) if x is not None else None
        h3, cl3 = ops.sru_compute(ug3, xg3, bg3, cg3, "tanh", 1.0)
        ((h3 * gh).sum() + (cl3 * gl).sum()).backward()
    finally:
        os.environ.pop("DMN_SRU_BWD_ATOMIC", None)
    assert rel_err(bg3.grad.cpu(), bg.grad.cpu()) < 1e-5


# ------------------------------------------------------------------------------------
# a11: the last scan stores its hidden states as the (B, Hm, H, W) map itself (dmn.py:209-212)
# ------------------------------------------------------------------------------------
@pytest.mark.parametrize("T,B,d,k,nchunks,strided", [
    (256, 32, 1005, 3, 1, True),      # the benchmarked tail: 503 waves, into the head's [x | feat] buffer
    (100, 64, 1005, 3, 1, False),     # 320 px: T not a multiple of the 64-step groups
    (256, 1, 1005, 3, 4, True),       # batch-1 serving: time-parallel variant
    (70, 3, 37, 4, 1, False),         # k = 4, waves straddling samples, ragged last wave
    (5, 2, 16, 3, 1, False),          # shorter than a group and than the ring
    (64, 1, 64, 3, 1, False), (65, 1, 63, 4, 2, True), (1, 1, 1, 3, 1, False), (130, 2, 130, 3, 3, False)])
@pytest.mark.parametrize("act", ["tanh", "relu"])
def test_sru_forward_image_is_bit_identical_to_scan_then_permute(T, B, d, k, nchunks, strided, act):
    u = det_uniform((T, B, d, k), 801, 1.5).to(DEV)
    x = det_uniform((T, B, d), 802, 1.0).to(DEV) if k == 3 else None
    bias = det_uniform((2 * d,), 803, 0.5).to(DEV)
    c0 = det_uniform((B, d), 804, 1.0).to(DEV)
    h, _, cl = ops.sru_forward_raw(u, x, bias, c0, act, 1.0, nchunks=nchunks)     # same chunking: same bits
    want = h.permute(1, 2, 0).contiguous()
    if strided:
        whole = torch.full((B, d + 7, T), -123.0, device=DEV)
        out, cl2 = ops.sru_forward_image(u, x, bias, c0, act, 1.0, out=whole[:, :d], nchunks=nchunks)
        assert out.data_ptr() == whole.data_ptr()
        assert bool((whole[:, d:] == -123.0).all())                  # nothing written past the map
    else:
        out, cl2 = ops.sru_forward_image(u, x, bias, c0, act, 1.0, nchunks=nchunks)
    assert out.shape == (B, d, T)
    assert torch.equal(out, want) and torch.equal(cl2, cl)


def test_hot_path_with_and_without_the_image_store_gives_the_same_bits_and_fills_the_head_buffer():
    """BaseDMN.hot_path (inference): scan_writes_image on / off must agree bit for bit (same kernels, same
    arithmetic, only the store differs), and with a destination the map must land in it."""
    cfg = dict(emb_size=32, hid_size=32, vis_size=64, num_filters=10, mixed_size=32, hid_mixed_size=37,
               lang_layers=2, mixed_layers=3, mix_we=True, high_res=True)
    net = dms_amd.BaseDMN(60, pretrained=False, **cfg)
    net.base = torch.nn.Identity()
    torch.nn.Module.load_state_dict(net, det_fill_state(net.state_dict()))
    net = net.to(DEV).eval()
    B, H, W, L = 3, 5, 4, 20
    vis = det_uniform((B, 64, H, W), 811, 1.0).clamp_(min=0).to(DEV)
    lang = det_ints((B, L), 812, 60).to(DEV)
    with torch.no_grad():
        net.scan_writes_image = False
        ref = net.hot_path(vis, lang)
        net.scan_writes_image = True
        got = net.hot_path(vis, lang)
        whole = torch.zeros((B, 37 + 11, H, W), device=DEV)
        got2 = net.hot_path(vis, lang, image_dest=whole[:, :37])
    assert ref.shape == (B, 37, H, W) and torch.equal(got, ref) and torch.equal(got2, ref)
    assert got2.data_ptr() == whole.data_ptr() and torch.equal(whole[:, :37], ref) and float(whole[:, 37:].abs().sum()) == 0.0


@pytest.mark.parametrize("rows,cols", [(640, 26960), (3, 7), (1, 4), (2050, 12), (20, 2696 * 10)])
def test_bias_sigmoid_in_place_matches_float64_linear_bias_plus_sigmoid(rows, cols):
    """dmn.py:173-174 (the Linear's bias add + F.sigmoid) as one in-place pass: vs float64."""
    x = det_uniform((rows, cols), 821, 6.0)
    b = det_uniform((cols,), 822, 1.0)
    got = ops.bias_sigmoid_(x.clone().to(DEV), b.to(DEV)).cpu()
    want = torch.sigmoid(x.double() + b.double())
    assert float((got.double() - want).abs().max()) < 3e-7       # sigmoid values are O(1): ~2 ulp
