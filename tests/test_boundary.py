"""CPU tests (-m "not gpu"): the C-ABI library loads and exports every declared symbol,
the module API mirrors the reference, and the product never reaches into oracle/."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import dms_amd
from dms_amd import _lib
from tests.util import load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "dmn_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dmn_[a-z0-9_]+)\s*\(", text)))


def test_library_loads_and_exports_every_declared_symbol():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = _declared_symbols()
    assert len(syms) >= 13
    for s in syms:
        assert hasattr(lib, s), "libdmn_hip.so does not export %s" % s
    lib.dmn_abi_version.restype = ctypes.c_int
    assert lib.dmn_abi_version() == _lib.ABI_VERSION
    # python-side signature table covers the header (minus the two no-arg helpers)
    assert set(_lib.SIGNATURES) | {"dmn_abi_version", "dmn_last_error"} == set(syms)


def test_bad_arguments_are_rejected_without_a_gpu():
    lib = _lib.load()
    # k = 5 is rejected before anything is launched (no device needed)
    rc = lib.dmn_sru_fwd(1, None, 1, None, 1, None, None, 4, 1, 8, 5, 1, 1.0, 0, None)
    assert rc < 0
    assert b"k must be 3 or 4" in lib.dmn_last_error()
    rc = lib.dmn_sru_fwd(1, None, 1, None, 1, None, None, 4, 1, 8, 3, 1, 1.0, 0, None)  # T=4
    assert rc < 0 and b"highway" in lib.dmn_last_error()
    with pytest.raises(RuntimeError, match="bad sizes"):
        _lib.call("dmn_dynfilter_fwd", 1, 1, 1, 0, 1, 1, 1, 1, None)


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libdmn_hip.so")
    with pytest.raises(RuntimeError, match="no fallback"):
        _lib.load()


def test_cpu_tensors_raise_no_cpu_fallback():
    net = dms_amd.BaseDMN(20, emb_size=8, hid_size=8, vis_size=16, num_filters=2, mixed_size=8,
                          hid_mixed_size=8, lang_layers=1, mixed_layers=1, pretrained=False)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net.hot_path(torch.zeros(1, 16, 2, 2), torch.zeros(1, 3, dtype=torch.long))


@pytest.mark.parametrize("hr", [True, False])
def test_state_dict_matches_reference(hr):
    g = load_golden("dmn_state_keys_highres" if hr else "dmn_state_keys_lowres")
    kw = dict(upsampling_mode="bilinear") if hr else {}
    net = dms_amd.DMN(100, num_filters=10, lang_layers=3, mix_we=True, pretrained=False,
                      high_res=hr, **kw)
    sd = net.state_dict()
    keys = sorted(sd)
    assert keys == list(g["keys"])
    assert [str(tuple(sd[k].shape)) for k in keys] == list(g["shapes"])


def test_ctor_defaults_match_reference_signature():
    import inspect
    sig = inspect.signature(dms_amd.DMN.__init__)
    d = {k: v.default for k, v in sig.parameters.items() if v.default is not inspect._empty}
    assert d == dict(emb_size=1000, hid_size=1000, vis_size=2688, num_filters=1, mixed_size=1000,
                     hid_mixed_size=1005, lang_layers=2, mixed_layers=3, backend='dpn92',
                     mix_we=False, lstm=False, pretrained=True, extra=True, high_res=False,
                     upsampling_mode='bilineal', upsampling_size=3,
                     upsampling_amplification=32, dmn_freeze=False)
    sig = inspect.signature(dms_amd.BaseDMN.__init__)
    d = {k: v.default for k, v in sig.parameters.items() if v.default is not inspect._empty}
    assert d == dict(emb_size=1000, hid_size=1000, vis_size=2688, num_filters=10,
                     mixed_size=1000, hid_mixed_size=1005, lang_layers=3, mixed_layers=3,
                     backend='dpn92', mix_we=True, lstm=False, pretrained=True, extra=True,
                     high_res=False)


def test_lenient_load_state_dict_copies_only_matching_sizes():
    kw = dict(emb_size=8, hid_size=8, vis_size=16, num_filters=2, mixed_size=8, lang_layers=1,
              mixed_layers=2, pretrained=False, mix_we=True)
    low = dms_amd.DMN(20, hid_mixed_size=8, high_res=False, **kw)
    other = dms_amd.DMN(20, hid_mixed_size=12, high_res=False, **kw)
    before = {k: v.clone() for k, v in other.state_dict().items()}
    other.load_state_dict(low.state_dict())
    after = other.state_dict()
    assert torch.equal(after["langvis.emb.weight"], low.state_dict()["langvis.emb.weight"])
    # different size (k=4, 12 wide): untouched
    k = "langvis.mrnn.rnn_lst.0.weight"
    assert after[k].shape != low.state_dict()[k].shape and torch.equal(after[k], before[k])
    # pretrained=True would download: refused
    with pytest.raises(RuntimeError, match="download"):
        dms_amd.DMN(20, pretrained=True)


def test_lenient_load_names_every_skipped_key_in_one_log_line(caplog):
    """SRU key names are recalled from the absent `sru` package, not verified: a checkpoint that
    names them differently must not load as random weights without a word."""
    import logging
    kw = dict(emb_size=8, hid_size=8, vis_size=16, num_filters=2, mixed_size=8, lang_layers=1,
              mixed_layers=2, pretrained=False, mix_we=True, hid_mixed_size=12)
    net = dms_amd.DMN(20, **kw)
    ckpt = {k: v.clone() for k, v in dms_amd.DMN(20, **kw).state_dict().items()}
    with caplog.at_level(logging.WARNING, logger="dms_amd"):
        assert net.load_state_dict(ckpt) == []                        # complete checkpoint: silent
    assert not caplog.records
    # a checkpoint whose SRU layers are named differently, one wrong size, and no BN counters
    renamed = {}
    for k, v in ckpt.items():
        if k.endswith("num_batches_tracked"):
            continue
        renamed[k.replace("mrnn.rnn_lst.0.weight", "mrnn.rnn_lst.0.W")] = v
    renamed["langvis.emb.weight"] = torch.zeros(21, 8)
    with caplog.at_level(logging.WARNING, logger="dms_amd"):
        skipped = net.load_state_dict(renamed)
    assert len(caplog.records) == 1
    msg = caplog.records[0].getMessage()
    assert "langvis.mrnn.rnn_lst.0.weight (absent)" in msg and "langvis.mrnn.rnn_lst.0.W" in msg
    assert "langvis.emb.weight (size (21, 8) != (20, 8))" in msg
    assert "num_batches_tracked" not in msg and "\n" not in msg
    assert "langvis.mrnn.rnn_lst.0.weight" in skipped and any(k.endswith("num_batches_tracked") for k in skipped)
    # BaseDMN's own loader reports as well
    caplog.clear()
    with caplog.at_level(logging.WARNING, logger="dms_amd"):
        net.langvis.load_state_dict({})
    assert len(caplog.records) == 1 and "emb.weight (absent)" in caplog.records[0].getMessage()


def test_product_never_imports_oracle_or_reference():
    pat = re.compile(r"^\s*(from|import)\s+(oracle|tests|dmn_pytorch|sru\b)", re.M)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "dms_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not pat.search(text), f
                assert "/root/reference" not in text, f


def test_sru_module_surface():
    rnn = dms_amd.SRU(6, 9, num_layers=3)
    shapes = {k: tuple(v.shape) for k, v in rnn.state_dict().items()}
    assert shapes == {"rnn_lst.0.weight": (6, 36), "rnn_lst.0.bias": (18,),
                      "rnn_lst.1.weight": (9, 27), "rnn_lst.1.bias": (18,),
                      "rnn_lst.2.weight": (9, 27), "rnn_lst.2.bias": (18,)}
    assert rnn.rnn_lst[0].k == 4 and rnn.rnn_lst[1].k == 3
    w = rnn.rnn_lst[0].weight
    assert float(w.abs().max()) <= (3.0 / 6) ** 0.5 + 1e-6 and float(rnn.rnn_lst[0].bias.abs().sum()) == 0
    with pytest.raises(NotImplementedError):
        dms_amd.SRU(4, 4, bidirectional=True)


def test_tap_decomposition_algebra_on_cpu():
    """The identity behind the head's tap GEMM (dms_amd/csrc/tapconv.hip), checked in plain torch:
    conv3x3_pad1(up2(X)) == sum over taps of shift_tap(up2(W_tap . X)) with zero padding at the
    HIGH-res border, W_tap rows laid out as ops.tap_weight does."""
    import torch.nn.functional as F
    from dms_amd import ops
    torch.manual_seed(0)
    B, C, Co, H, W = 2, 5, 3, 4, 6
    x = torch.randn(B, C, H, W, dtype=torch.float64)
    w = torch.randn(Co, C, 3, 3, dtype=torch.float64)
    want = F.conv2d(F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False), w, padding=1)
    wt = ops.tap_weight(w)                                       # (9*Co, C), row = tap*Co + o
    assert wt.shape == (9 * Co, C)
    Z = torch.einsum("rc,bchw->brhw", wt, x).view(B, 9, Co, H, W)
    got = torch.zeros_like(want)
    for ky in range(3):
        for kx in range(3):
            up = F.interpolate(Z[:, ky * 3 + kx], scale_factor=2, mode="bilinear", align_corners=False)
            up = F.pad(up, (1, 1, 1, 1))                          # zero outside the high-res image
            got += up[:, :, ky:ky + 2 * H, kx:kx + 2 * W]
    assert float((got - want).abs().max()) < 1e-12


def test_dmn_links_backbone_taps_to_head_levels():
    """DMN tells the backbone how many channels each head level puts in front of its pyramid tap
    (cat([x, features[i]]) of dmn.py:337 without a copy): level i reads features[4 - i]."""
    import dms_amd
    net = dms_amd.DMN(50, num_filters=10, lang_layers=3, mix_we=True, pretrained=False, high_res=True,
                      upsampling_mode="bilinear", upsampling_amplification=32)
    assert net.langvis.base.tap_prefix == {"conv5_3": 1005, "conv4_20": 512, "conv3_4": 256, "conv2_3": 128}
    low = dms_amd.DMN(50, num_filters=10, lang_layers=3, mix_we=True, pretrained=False, high_res=False)
    assert low.langvis.base.tap_prefix is None


def test_bench_gpus_n_without_torchrun_refuses_a_box_with_fewer_devices():
    """`python bench.py --gpus 2` (the driver's form: no torchrun environment) must start its ranks itself or
    fail loudly -- never run one rank and print a line.  This container has no GPU: it must refuse, non-zero,
    naming the device count, before any rank is started."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "DMN_BENCH_DEVICE")}
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer than 2 GPUs")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode != 0
    assert b"GPU(s)" in r.stderr and b"refusing" in r.stderr
    assert not [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]


def test_committed_bench_line_keeps_the_driver_contract():
    """profiles/r3_bench_n1.json is the line `python bench.py` printed on the MI355X box: the keys the driver and the
    judge read must all be there, with the relations the contract states (frac = achieved / peak, value * ms_per_step
    = batch, the CPU leg says what it ran on)."""
    import json
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r3_bench_n1.json")
    line = json.loads(open(path).read().strip().splitlines()[-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["higher_is_better"] is True and line["scaling"] == "weak"
    assert line["vs_baseline"] is None and line["dtype"] == "f32" and line["data"] == "synthetic"
    assert "workload" in line["config"] and "model" not in line["config"]
    batch = line["config"]["global_batch"]
    assert abs(line["value"] * line["ms_per_step"] / 1e3 - batch) < 0.01 * batch
    r = line["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0.0 < r["frac"] < 1.0
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 0.01 * r["achieved"]
    assert r["traffic"] is None or 0.99 < r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.2
    c = line["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    assert line["collective_ranks"] == 1
    for shard in ("B=64", "B=128"):
        assert line["shards"][shard]["finite"] and line["shards"][shard]["rel_diff_vs_same_samples_in_B32"] < 1e-4
    assert line["train320"]["value"] > 0 and line["train320"]["batch_per_gpu"] == 64
