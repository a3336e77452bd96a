"""Shared helpers for the tests (oracle-side; never imported by dms_amd)."""
import os

import numpy as np
import torch

from oracle.detgen import det_fill_state, det_ints, det_uniform

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def cfg_from_golden(g):
    cfg = {}
    for k in g.files:
        if k.startswith("cfg_"):
            v = g[k].item()
            cfg[k[4:]] = v
    return cfg


def rel_err(a, b):
    """max |a-b| / max |b|  (the 'relative in fp32' measure used for logits)."""
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def assert_close_elementwise(a, b, rtol=1e-4, atol_rel=1e-6):
    """Element-wise companion of rel_err: |a - b| <= rtol * |b| + atol_rel * max|b| for EVERY element
    (north_star: 'within 1e-4 relative in fp32'); the absolute term keeps logits that are zero to
    rounding from needing an exact match, and is 100x tighter than the norm-wise bound."""
    a = np.asarray(torch.as_tensor(a).detach().cpu().double())
    b = np.asarray(torch.as_tensor(b).detach().cpu().double())
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol_rel * float(np.abs(b).max()))


def base_inputs(cfg, H, W, L, dict_size):
    vis = det_uniform((1, cfg["vis_size"], H, W), 101, 1.0).clamp_(min=0)
    lang = det_ints((1, L), 202, dict_size)
    return vis, lang
