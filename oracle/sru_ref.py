# -*- coding: utf-8 -*-
"""oracle/sru_ref.py -- TEST INFRASTRUCTURE ONLY (not product code).

Two CPU restatements of the SRU v1 recurrence used by the reference through its
un-vendored third-party `sru` package (call sites dmn_pytorch/models/dmn.py:7,104,
120,156,204; pin: README.md:33 `taolei87/sru@43c85ed`):

* ``SRURef`` / ``SRUCellRef``: pure-torch modules (python loop over time) with the
  public surface the reference uses -- ``SRU(input_size, hidden_size, num_layers=n)``
  and ``h, c = rnn(x)`` with ``x (len, B, n_in)``.  Autograd of these equations is
  the independent check of the hand-written backward passes (C oracle and HIP).
* ctypes bindings of ``oracle/sru_oracle.c`` (plain C, optional OpenMP), used for
  fast CPU checking at the full sizes and as ``bench.py``'s ``cpu_baseline``.

PARITY UNPINNED for the SRU arithmetic itself: the `sru` source is absent from
/root/reference and from this container and the reference holds no golden vector
for it.  The equations follow Lei et al. arXiv:1709.02755 in the per-column kernel
form recorded in SURVEY.md section 2.1; ``activation`` and ``scale_x`` are explicit knobs.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes
import math
import os
import subprocess

import numpy as np
import torch
import torch.nn as nn

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libsru_oracle.so")
_SRC = os.path.join(_HERE, "sru_oracle.c")

ACT_IDS = {"identity": 0, "tanh": 1, "relu": 2}


def build(force=False):
    """Compile oracle/sru_oracle.c -> oracle/_build/libsru_oracle.so (gcc, -fopenmp)."""
    if (not force and os.path.exists(_SO)
            and os.path.getmtime(_SO) >= os.path.getmtime(_SRC)):
        return _SO
    os.makedirs(os.path.dirname(_SO), exist_ok=True)
    # -ffp-contract=off: no fused multiply-add, so the C oracle rounds like the torch one
    cmd = ["gcc", "-O2", "-ffp-contract=off", "-fopenmp", "-shared", "-fPIC",
           "-o", _SO, _SRC, "-lm"]
    subprocess.run(cmd, check=True)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        fp = ctypes.POINTER(ctypes.c_float)
        L.sru_oracle_fwd.argtypes = [fp, fp, fp, fp, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_float, fp, fp, fp]
        L.sru_oracle_fwd.restype = ctypes.c_int
        L.sru_oracle_bwd.argtypes = [fp, fp, fp, fp, fp, fp, fp, ctypes.c_int,
                                     ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_int, ctypes.c_float, fp, fp, fp, fp]
        L.sru_oracle_bwd.restype = ctypes.c_int
        L.sru_oracle_set_threads.argtypes = [ctypes.c_int]
        L.sru_oracle_set_threads.restype = ctypes.c_int
        _lib = L
    return _lib


def _p(a):
    if a is None:
        return None
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def set_threads(n):
    return lib().sru_oracle_set_threads(int(n))


def sru_fwd_c(u, x, bias, c0=None, act="tanh", scale_x=1.0, want_c=True):
    """u (T,B,d,k) f32; x (T,B,d) or None (k==4); bias (2d,). -> h, c_all|None, c_last."""
    T, B, d, k = u.shape
    h = np.empty((T, B, d), np.float32)
    c = np.empty((T, B, d), np.float32) if want_c else None
    c_last = np.empty((B, d), np.float32)
    rc = lib().sru_oracle_fwd(_p(u), _p(x), _p(bias), _p(c0), T, B, d, k,
                              ACT_IDS[act], float(scale_x), _p(h), _p(c), _p(c_last))
    if rc != 0:
        raise RuntimeError("sru_oracle_fwd rc=%d" % rc)
    return h, c, c_last


def sru_bwd_c(u, x, bias, c0, c, grad_h, grad_last, act="tanh", scale_x=1.0):
    T, B, d, k = u.shape
    gu = np.empty_like(u)
    gx = np.empty((T, B, d), np.float32) if k == 3 else None
    gb = np.empty((2, B, d), np.float32)
    gc0 = np.empty((B, d), np.float32)
    rc = lib().sru_oracle_bwd(_p(u), _p(x), _p(bias), _p(c0), _p(c), _p(grad_h),
                              _p(grad_last), T, B, d, k, ACT_IDS[act], float(scale_x),
                              _p(gu), _p(gx), _p(gb), _p(gc0))
    if rc != 0:
        raise RuntimeError("sru_oracle_bwd rc=%d" % rc)
    return gu, gx, gb, gc0


# ----------------------------------------------------------------------------------
# pure-torch restatement (differentiable)
# ----------------------------------------------------------------------------------
def _act(name, v):
    if name == "tanh":
        return torch.tanh(v)
    if name == "relu":
        return torch.relu(v)
    return v


def sru_scan_torch(u, x, bias, c0=None, act="tanh", scale_x=1.0):
    """u (T,B,d,k) ; x (T,B,d) or None ; bias (2d,) -> h (T,B,d), c_all (T,B,d)."""
    T, B, d, k = u.shape
    bf, br = bias[:d], bias[d:]
    f = torch.sigmoid(u[..., 1] + bf)
    r = torch.sigmoid(u[..., 2] + br)
    u0 = u[..., 0]
    xp = x * scale_x if k == 3 else u[..., 3]
    cur = u.new_zeros(B, d) if c0 is None else c0
    cs = []
    for t in range(T):
        cur = (cur - u0[t]) * f[t] + u0[t]
        cs.append(cur)
    c = torch.stack(cs) if T > 0 else u.new_zeros(0, B, d)
    h = (_act(act, c) - xp) * r + xp
    return h, c


class SRUCellRef(nn.Module):
    """One SRU layer; parameter names/shapes as recalled from the `sru` package:
    ``weight (n_in, n_out*k)`` with k innermost, ``bias (2*n_out)`` = [forget | reset]."""

    def __init__(self, n_in, n_out, activation="tanh", scale_x=1.0):
        super().__init__()
        self.n_in, self.n_out = n_in, n_out
        self.k = 4 if n_in != n_out else 3
        self.activation = activation
        self.scale_x = scale_x
        self.weight = nn.Parameter(torch.empty(n_in, n_out * self.k))
        self.bias = nn.Parameter(torch.empty(n_out * 2))
        val = math.sqrt(3.0 / n_in)
        with torch.no_grad():
            self.weight.uniform_(-val, val)
            self.bias.zero_()

    def forward(self, x, c0=None):
        T, B, _ = x.shape
        u = x.contiguous().view(-1, self.n_in).mm(self.weight)
        u = u.view(T, B, self.n_out, self.k)
        h, c = sru_scan_torch(u, x if self.k == 3 else None, self.bias, c0,
                              self.activation, self.scale_x)
        return h, c[-1]


class SRURef(nn.Module):
    """Stack of SRU layers: ``SRU(input_size, hidden_size, num_layers=2)``;
    ``forward(x[, c0]) -> (h (len,B,hidden), c (num_layers,B,hidden))``."""

    def __init__(self, input_size, hidden_size, num_layers=2, activation="tanh",
                 scale_x=1.0, **unused):
        super().__init__()
        self.n_in, self.n_out, self.depth = input_size, hidden_size, num_layers
        self.rnn_lst = nn.ModuleList([
            SRUCellRef(input_size if i == 0 else hidden_size, hidden_size,
                       activation, scale_x) for i in range(num_layers)])

    def forward(self, x, c0=None):
        assert x.dim() == 3
        c0s = [None] * self.depth if c0 is None else list(c0.unbind(0))
        prev, lst = x, []
        for cell, ci in zip(self.rnn_lst, c0s):
            prev, c = cell(prev, ci)
            lst.append(c)
        return prev, torch.stack(lst)
