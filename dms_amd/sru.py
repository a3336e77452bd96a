# -*- coding: utf-8 -*-
"""SRU modules backed by the HIP scan kernels (drop-in for ``from sru import SRU``).

The reference builds its language encoder and its multimodal RNN with the third-party
package ``sru`` (dmn_pytorch/models/dmn.py:7,104,120) and calls ``h, c = rnn(x)`` with
``x (len, B, n_in)`` (dmn.py:156,204).  This module keeps that surface -- constructor
``SRU(input_size, hidden_size, num_layers=2, ...)``, ``forward(input, c0=None)`` returning
``(h (len,B,hidden), c (num_layers,B,hidden))``, parameters ``rnn_lst.{i}.weight
(n_in, hidden*k)`` (k innermost; k = 3 when n_in == hidden else 4) and ``rnn_lst.{i}.bias
(2*hidden)`` = [forget | reset] -- and replaces the cupy/pynvrtc CUDA kernel by
libdmn_hip.so.  The x.W projection stays a library GEMM (rocBLAS/hipBLASLt via torch).

Not implemented (unused by the reference, which constructs SRU with defaults only):
dropout / rnn_dropout, bidirectional, layer/weight norm.
"""
import math

import torch
import torch.nn as nn

from . import ops


class SRUCell(nn.Module):
    def __init__(self, n_in, n_out, use_tanh=1, use_relu=0, scale_x=1.0):
        super().__init__()
        self.n_in, self.n_out = n_in, n_out
        self.k = 4 if n_in != n_out else 3
        self.activation = "relu" if use_relu else ("tanh" if use_tanh else "identity")
        self.scale_x = float(scale_x)
        self.weight = nn.Parameter(torch.empty(n_in, n_out * self.k))
        self.bias = nn.Parameter(torch.empty(n_out * 2))
        self.state_only_head = True     # inference shortcut of _forward_tail; False = one full scan
        self.state_only_head_in_training = True   # the same under autograd (round 2)
        self.init_weight()

    def init_weight(self):
        val = math.sqrt(3.0 / self.n_in)
        with torch.no_grad():
            self.weight.uniform_(-val, val)
            self.bias.zero_()

    def set_bias(self, bias_val=0.0):
        with torch.no_grad():
            self.bias[self.n_out:].fill_(bias_val)

    def forward_projected(self, u, c0=None, h_t0=0):
        """Scan on an already projected input ``u (len, B, n_out, 4)`` (k = 4 cells only: the
        highway term is U[..., 3], so the raw input is not needed).  Used when the caller has
        folded this cell's x.W into an upstream linear map."""
        if self.k != 4:
            raise ValueError("forward_projected needs a k = 4 cell (n_in != n_out)")
        T, B = u.shape[0], u.shape[1]
        return ops.sru_compute(u.reshape(T, B, self.n_out, 4), None, self.bias, c0,
                               self.activation, self.scale_x, h_t0)

    def _state_weight(self):
        """The [u0 | forget] columns of ``weight`` as (n_in, n_out*2), cached per weight version."""
        w = self.weight
        key = (w._version, w.data_ptr())
        cached = self.__dict__.get("_w01")
        if cached is None or cached[0] != key:
            with torch.no_grad():
                w01 = w.detach().view(self.n_in, self.n_out, self.k)[:, :, :2].reshape(self.n_in, -1)
                cached = (key, w01.contiguous())
            self.__dict__["_w01"] = cached
        return cached[1]

    def _forward_tail(self, x, c0, h_t0, image_out=None):
        """Inference, hidden states wanted from step h_t0 on only: the earlier steps reach the
        output through the cell state alone, so they are projected onto the u0 / forget
        columns (2 of k: a third less GEMM work for k = 3) and run through the state-only scan.
        image_out: True or a (B, n_out, T - h_t0) batch-strided view -> the hidden states come back
        as that image, (B, n_out, T - h_t0), instead of (T - h_t0, B, n_out)."""
        T, B, n_in = x.shape
        head = x[:h_t0].reshape(h_t0 * B, n_in)
        u2 = ops.tuned_matmul(head, self._state_weight()).view(h_t0, B, self.n_out, 2)
        c_mid = ops.sru_state(u2, self.bias, c0)
        tail = x[h_t0:]
        u = ops.tuned_matmul(tail.reshape((T - h_t0) * B, n_in), self.weight).view(
            T - h_t0, B, self.n_out, self.k)
        if image_out is not None:
            return ops.sru_forward_image(u, tail if self.k == 3 else None, self.bias, c_mid, self.activation,
                                         self.scale_x, out=None if image_out is True else image_out)
        return ops.sru_compute(u, tail if self.k == 3 else None, self.bias, c_mid,
                               self.activation, self.scale_x, 0)

    def forward(self, input, c0=None, h_t0=0, image_out=None):
        if input.dim() != 3:
            raise ValueError("SRUCell expects (len, batch, n_in), got %s" % (tuple(input.shape),))
        T, B, n_in = input.shape
        if n_in != self.n_in:
            raise ValueError("SRUCell: input width %d != n_in %d" % (n_in, self.n_in))
        x = input.contiguous()
        needs_grad = torch.is_grad_enabled() and (
            x.requires_grad or self.weight.requires_grad or self.bias.requires_grad
            or (c0 is not None and c0.requires_grad))
        if image_out is not None and (needs_grad or not (0 < h_t0 < T and self.state_only_head)):
            raise RuntimeError("SRUCell: image_out needs inference mode, 0 < h_t0 < T and state_only_head")
        if 0 < h_t0 < T and not needs_grad and self.state_only_head:
            return self._forward_tail(x, c0, h_t0, image_out)
        if needs_grad and 0 < h_t0 < T and self.state_only_head_in_training:
            # training, hidden states wanted from step h_t0 on only: the earlier steps reach the loss
            # through the cell state alone, so they -- and their dgrad / wgrad GEMMs -- run on the
            # [u0 | forget] weight columns only (2 of k), with the state-only scan's own backward
            w01 = self.weight.view(self.n_in, self.n_out, self.k)[:, :, :2].reshape(self.n_in, -1)
            u2 = x[:h_t0].reshape(h_t0 * B, n_in).mm(w01).view(h_t0, B, self.n_out, 2)
            c_mid = ops.sru_state_train(u2, self.bias, c0)
            tail = x[h_t0:]
            u = tail.reshape((T - h_t0) * B, n_in).mm(self.weight).view(T - h_t0, B, self.n_out, self.k)
            return ops.sru_compute(u, tail if self.k == 3 else None, self.bias, c_mid,
                                   self.activation, self.scale_x, 0)
        if needs_grad:
            u = x.view(T * B, n_in).mm(self.weight)
        else:                                   # inference: BLAS library chosen per shape
            u = ops.tuned_matmul(x.view(T * B, n_in), self.weight)
        u = u.view(T, B, self.n_out, self.k)
        return ops.sru_compute(u, x if self.k == 3 else None, self.bias, c0,
                               self.activation, self.scale_x, h_t0)


class SRU(nn.Module):
    def __init__(self, input_size, hidden_size, num_layers=2, dropout=0, rnn_dropout=0,
                 bidirectional=False, use_tanh=1, use_relu=0, scale_x=1.0):
        super().__init__()
        if dropout or rnn_dropout or bidirectional:
            raise NotImplementedError(
                "dms_amd.SRU: dropout / bidirectional are not on the DMN path (dmn.py:104,120 "
                "construct SRU with defaults) and are not implemented")
        self.n_in, self.n_out, self.depth = input_size, hidden_size, num_layers
        self.rnn_lst = nn.ModuleList([
            SRUCell(input_size if i == 0 else hidden_size, hidden_size, use_tanh, use_relu, scale_x)
            for i in range(num_layers)])

    def set_bias(self, bias_val=0.0):
        for cell in self.rnn_lst:
            cell.set_bias(bias_val)

    def can_image(self, T, tail, layer0_external=False):
        """Whether forward(..., tail=tail, image_out=...) is available: inference, a last layer that is not
        a layer 0 run or projected by the caller, a real head (0 < T - tail) and the state-only head enabled."""
        if torch.is_grad_enabled() or tail is None or not (0 < T - int(tail) < T):
            return False
        if self.depth == 1 and layer0_external:
            return False
        return bool(self.rnn_lst[-1].state_only_head)

    def forward(self, input, c0=None, return_hidden=True, tail=None, projected=False, first_done=None,
                image_out=None):
        """``tail=n`` (extension): only the last n steps of the LAST layer's h are produced
        (what dmn.py:209 keeps), so the final scan stores S instead of S*L rows.
        ``projected=True`` (extension): ``input`` is already layer 0's U = x.W,
        shape (len, B, hidden*4); layer 0 must be a k = 4 cell.
        ``first_done=c_last`` (extension): ``input`` is already layer 0's hidden sequence (the
        caller ran layer 0 itself, e.g. ops.mix_sru_fwd) and c_last its final cell state.
        ``image_out`` (extension, inference, with ``tail``): True or a (B, hidden, tail) batch-strided
        view -> the last layer's kept hidden states come back as that image -- what dmn.py:210-212
        builds with permute(1, 2, 0) -- written by the scan itself; see can_image()."""
        if input.dim() != 3:
            raise ValueError("SRU expects (len, batch, n_in), got %s" % (tuple(input.shape),))
        T = input.shape[0]
        c0s = [None] * self.depth if c0 is None else [c.contiguous() for c in c0.unbind(0)]
        prev, lst = input, []
        for i, cell in enumerate(self.rnn_lst):
            last = i == self.depth - 1
            h_t0 = max(T - int(tail), 0) if (last and tail is not None) else 0
            if i == 0 and first_done is not None:
                c = first_done
                if last and h_t0:
                    prev = prev[h_t0:]
            elif i == 0 and projected:
                prev, c = cell.forward_projected(prev, c0s[i], h_t0)
            elif last and image_out is not None:
                prev, c = cell(prev, c0s[i], h_t0, image_out=image_out)
            else:
                prev, c = cell(prev, c0s[i], h_t0)
            lst.append(c)
        if return_hidden:
            return prev, torch.stack(lst)
        return prev
