# -*- coding: utf-8 -*-
"""Dual Path Network feature extractor (stays on MIOpen; out of the HIP hot path).

Own definition of the backbone the reference builds with ``create_model('dpn92', 1,
pretrained=..., extra=...)`` (dmn_pytorch/models/dmn.py:96-98 ->
dmn_pytorch/models/dpn/dpn_model.py:272-374).  What must match for drop-in use is the
state_dict (names + shapes, so published DMN checkpoints load) and the five pyramid taps:

* features[0]: the raw 7x7/2 stem convolution BEFORE batch-norm (dpn_model.py:191-197;
  the reference evaluates that convolution twice, here once -- same values);
* features[1..4]: cat(residual, dense) after conv2_3 / conv3_4 / conv4_20 / conv5_3
  (dpn_model.py:358-362); output = conv5_bn_ac of the last block (2688 ch for dpn92).

No torchvision import and no URL download: ``pretrained=True`` raises.

Inference fast path (eval mode, grad disabled, fp32 on a ROCm device): the element-wise glue
between the MIOpen convolutions -- BatchNorm + ReLU, the dual-path add / cat and the next
block's cat -- runs in two HIP kernels (csrc/dpn_glue.hip) instead of ~6 torch kernels per
block; a block's output is ONE contiguous [residual | dense] tensor.  Same values (eval-mode
BN is an affine map; differences are fp32 rounding, ~1e-7 relative).
"""
from collections import OrderedDict

import torch
import torch.nn as nn

from .. import ops

BN_EPS = 1e-3

#          stem  k_r  groups  blocks per stage   dense increment     small  b
VARIANTS = {
    "dpn68": (10, 128, 32, (3, 4, 12, 3), (16, 32, 32, 64), True, False),
    "dpn68b": (10, 128, 32, (3, 4, 12, 3), (16, 32, 32, 64), True, True),
    "dpn92": (64, 96, 32, (3, 4, 20, 3), (16, 32, 24, 128), False, False),
    "dpn98": (96, 160, 40, (3, 6, 20, 3), (16, 32, 32, 128), False, False),
    "dpn131": (128, 160, 40, (4, 8, 28, 3), (16, 32, 32, 128), False, False),
    "dpn107": (128, 200, 50, (4, 8, 20, 3), (20, 64, 64, 128), False, False),
}


def _join(x):
    return torch.cat(x, dim=1) if isinstance(x, tuple) else x


def _bn_consts(bn):
    """(mean, scale, bias) of an eval-mode BatchNorm2d, cached until its tensors change."""
    tensors = (bn.weight, bn.bias, bn.running_mean, bn.running_var)
    key = tuple((t._version, t.data_ptr()) for t in tensors)
    cached = bn.__dict__.get("_dmn_consts")
    if cached is None or cached[0] != key:
        with torch.no_grad():
            scale = (bn.weight * torch.rsqrt(bn.running_var + bn.eps)).contiguous()
            cached = (key, (bn.running_mean.detach().contiguous(), scale,
                            bn.bias.detach().contiguous()))
        bn.__dict__["_dmn_consts"] = cached
    return cached[1]


def _gconv_weight(conv):
    """Grouped conv weight in the kernel's (groups, ci, 9, co) layout, cached per weight version."""
    w = conv.weight
    key = (w._version, w.data_ptr())
    cached = conv.__dict__.get("_dmn_gconv_w")
    if cached is None or cached[0] != key:
        cached = (key, ops.gconv_weight_layout(w, conv.groups))
        conv.__dict__["_dmn_gconv_w"] = cached
    return cached[1]


def _proj_weight(conv, c1c):
    """Projection weight (c1c + 2*inc, Cin) with its rows reordered to [dense0 | res]: the order
    of the stage buffer (see ops.dpn_stage_pass); cached per weight version."""
    w = conv.weight
    key = (w._version, w.data_ptr(), c1c)
    cached = conv.__dict__.get("_dmn_proj_w")
    if cached is None or cached[0] != key:
        with torch.no_grad():
            w2 = w.detach().view(w.shape[0], w.shape[1])
            cached = (key, torch.cat([w2[c1c:], w2[:c1c]], 0).contiguous())
        conv.__dict__["_dmn_proj_w"] = cached
    return cached[1]


def _bn_relu(bn, x):
    return ops.bn_relu(x, *_bn_consts(bn))


def _conv(conv, x):
    """Inference convolution: 1x1 / stride 1 goes to the per-shape tuned batched GEMM."""
    if (conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.groups == 1
            and conv.padding == (0, 0) and conv.bias is None):
        return ops.conv1x1(x, conv.weight)
    return conv(x)


class CatBnAct(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.bn = nn.BatchNorm2d(ch, eps=BN_EPS)
        self.act = nn.ReLU(inplace=True)

    def forward(self, x):
        return self.act(self.bn(_join(x)))


class BnActConv2d(nn.Module):
    def __init__(self, cin, cout, kernel, stride, padding=0, groups=1):
        super().__init__()
        self.bn = nn.BatchNorm2d(cin, eps=BN_EPS)
        self.act = nn.ReLU(inplace=True)
        self.conv = nn.Conv2d(cin, cout, kernel, stride, padding, groups=groups, bias=False)

    def forward(self, x):
        return self.conv(self.act(self.bn(x)))


class InputBlock(nn.Module):
    def __init__(self, ch, kernel=7, padding=3):
        super().__init__()
        self.conv = nn.Conv2d(3, ch, kernel, stride=2, padding=padding, bias=False)
        self.bn = nn.BatchNorm2d(ch, eps=BN_EPS)
        self.act = nn.ReLU(inplace=True)
        self.pool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)

    def forward(self, x):
        raw = self.conv(x)
        # bn must not run in place on `raw`: it is returned as pyramid level 0
        return self.pool(self.act(self.bn(raw))), raw


class DualPathBlock(nn.Module):
    """kind: 'proj' (stride 1 projection), 'down' (stride 2 projection) or 'normal'."""

    def __init__(self, cin, c1a, c3b, c1c, inc, groups, kind="normal", b=False):
        super().__init__()
        self.c1c, self.inc, self.b = c1c, inc, b
        self.kind = kind
        self.fuse_gconv = True     # inference: own grouped 3x3 kernel with both BN+ReLU fused
        stride = 2 if kind == "down" else 1
        if kind == "down":
            self.c1x1_w_s2 = BnActConv2d(cin, c1c + 2 * inc, 1, 2)
        elif kind == "proj":
            self.c1x1_w_s1 = BnActConv2d(cin, c1c + 2 * inc, 1, 1)
        self.c1x1_a = BnActConv2d(cin, c1a, 1, 1)
        self.c3x3_b = BnActConv2d(c1a, c3b, 3, stride, padding=1, groups=groups)
        if b:
            self.c1x1_c = CatBnAct(c3b)
            self.c1x1_c1 = nn.Conv2d(c3b, c1c, 1, bias=False)
            self.c1x1_c2 = nn.Conv2d(c3b, inc, 1, bias=False)
        else:
            self.c1x1_c = BnActConv2d(c3b, c1c + inc, 1, 1)

    def forward(self, x):
        xin = _join(x)
        if self.kind == "normal":
            res, dense = x
        else:
            proj = self.c1x1_w_s2(xin) if self.kind == "down" else self.c1x1_w_s1(xin)
            res, dense = proj[:, :self.c1c], proj[:, self.c1c:]
        y = self.c3x3_b(self.c1x1_a(xin))
        if self.b:
            y = self.c1x1_c(y)
            o1, o2 = self.c1x1_c1(y), self.c1x1_c2(y)
        else:
            y = self.c1x1_c(y)
            o1, o2 = y[:, :self.c1c], y[:, self.c1c:]
        return res + o1, torch.cat([dense, o2], dim=1)

    def _y_path(self, act_a):
        """relu(c1x1_c.bn(c3x3_b(c1x1_a.conv(act_a)))): everything of the block's main path up to
        the input of its last 1x1 convolution."""
        y = _conv(self.c1x1_a.conv, act_a)
        conv_b = self.c3x3_b.conv
        stride = conv_b.stride[0]
        if (self.fuse_gconv and conv_b.in_channels // conv_b.groups in ops.GCONV_GROUP_WIDTHS
                and (stride == 1 or (stride == 2 and y.shape[2] % 2 == 0 and y.shape[3] % 8 == 0))):
            # bn + relu, grouped 3x3 conv, next bn + relu: one HBM-bound pass
            y = ops.gconv3x3_bn_relu(y, _gconv_weight(conv_b), conv_b.groups,
                                     _bn_consts(self.c3x3_b.bn), _bn_consts(self.c1x1_c.bn), stride)
        else:
            y = _bn_relu(self.c1x1_c.bn, conv_b(_bn_relu(self.c3x3_b.bn, y)))
        return y

    def forward_fused(self, X, act_a, next_bn):
        """X: contiguous [residual | dense]; act_a: relu(c1x1_a.bn(X)) if the previous merge
        already produced it.  Returns (X', relu(next_bn(X')))."""
        if act_a is None:
            act_a = _bn_relu(self.c1x1_a.bn, X)
        if self.kind == "normal":
            base = X
        else:
            w = self.c1x1_w_s2 if self.kind == "down" else self.c1x1_w_s1
            if (self.kind == "down" and X.shape[2] % 2 == 0 and X.shape[3] % 4 == 0
                    and w.conv.kernel_size == (1, 1) and w.conv.stride == (2, 2)):
                # a stride-2 1x1 conv reads a quarter of its input: activate only that quarter,
                # then it is the same batched GEMM as the stride-1 case
                base = ops.conv1x1(ops.bn_relu_s2(X, *_bn_consts(w.bn)), w.conv.weight)
            else:
                base = _conv(w.conv, _bn_relu(w.bn, X))
        y = _conv(self.c1x1_c.conv, self._y_path(act_a))
        return ops.dpn_merge(base, y, self.c1c, _bn_consts(next_bn))


class DPN(nn.Module):
    def __init__(self, variant="dpn92"):
        super().__init__()
        stem, k_r, groups, k_sec, inc_sec, small, b = VARIANTS[variant]
        bw_factor = 1 if small else 4
        blocks = OrderedDict()
        blocks["conv1_1"] = InputBlock(stem, 3 if small else 7, 1 if small else 3)
        cin = stem
        self.taps = []
        for stage in range(4):
            bw = 64 * (2 ** stage) * bw_factor
            inc = inc_sec[stage]
            r = (k_r * bw) // (64 * bw_factor)
            name = "conv%d_" % (stage + 2)
            blocks[name + "1"] = DualPathBlock(cin, r, r, bw, inc, groups,
                                               "proj" if stage == 0 else "down", b)
            cin = bw + 3 * inc
            for i in range(2, k_sec[stage] + 1):
                blocks[name + str(i)] = DualPathBlock(cin, r, r, bw, inc, groups, "normal", b)
                cin += inc
            self.taps.append(name + str(k_sec[stage]))
        blocks["conv5_bn_ac"] = CatBnAct(cin)
        self.out_channels = cin
        self.features = nn.Sequential(blocks)
        self.fusable = not b
        self.fuse_glue = True          # set False to force the plain torch path
        self.stage_inplace = True      # fused path: one in-place buffer per stage (False: dpn_merge per block)
        # fused path: {tap name: p}: that pyramid tap is written as the LAST channels of a
        # (N, p + C, H, W) buffer (returned as the view buf[:, p:]) so that a consumer can put p
        # channels of its own in front and read [its | tap] as one tensor (the high-res head's
        # per-level GEMM; set by DMN).  None / missing name: compact tap.
        self.tap_prefix = None

    def _forward_fused(self, x):
        children = list(self.features.named_children())
        stem = children[0][1]
        blocks, final = children[1:-1], children[-1][1]
        raw = stem.conv(x)
        pool = stem.pool
        if (raw.shape[3] % 4 == 0 and pool.kernel_size == 3 and pool.stride == 2 and pool.padding == 1
                and pool.dilation == 1 and not pool.ceil_mode):
            X = ops.bn_relu_maxpool(raw, *_bn_consts(stem.bn))    # bn + relu + pool: one pass
        else:
            X = pool(_bn_relu(stem.bn, raw))
        feats = [raw]
        act = None
        i = 0
        while i < len(blocks):
            j = i + 1
            while j < len(blocks) and blocks[j][1].kind == "normal":
                j += 1
            nxt = blocks[j][1].c1x1_a.bn if j < len(blocks) else final.bn
            if self.stage_inplace and self._stage_ok(blocks[i:j], X):
                X, act = self._run_stage(blocks[i:j], X, act, nxt)
                feats.append(X)
            else:
                X = X.contiguous()                             # (a prefixed tap is a batch-strided view)
                for k in range(i, j):
                    name, block = blocks[k]
                    nb = blocks[k + 1][1].c1x1_a.bn if k + 1 < len(blocks) else final.bn
                    X, act = block.forward_fused(X, act, nb)
                    if name in self.taps:
                        feats.append(X)
            i = j
        return act, feats

    def _stage_ok(self, stage, X):
        """The in-place stage path needs: a projection block first, its output being a pyramid tap
        (so only the stage's last block output is ever needed in reference order), plain 1x1
        convolutions, and for a stride-2 projection the sizes dmn_bn_relu_s2 takes."""
        first, last_name = stage[0][1], stage[-1][0]
        if first.kind == "normal" or last_name not in self.taps:
            return False
        if any(n in self.taps for n, _ in stage[:-1]):
            return False
        w = first.c1x1_w_s2 if first.kind == "down" else first.c1x1_w_s1
        if w.conv.kernel_size != (1, 1) or w.conv.bias is not None:
            return False
        if first.kind == "down" and (X.shape[2] % 2 or X.shape[3] % 4 or w.conv.stride != (2, 2)):
            return False
        return all(b.c1c == first.c1c and b.inc == first.inc and b.c1x1_c.conv.bias is None
                   for _, b in stage)

    def _run_stage(self, stage, X, act, next_bn):
        """One DPN stage on ONE buffer per sample, rows [dense0 | res | slot | o2_1 | ...]: the
        projection GEMM writes [dense0 | res], every block's last 1x1 convolution is a GEMM with
        beta = 1 onto [res | slot] (the residual add is its epilogue) and ops.dpn_stage_pass writes
        the next pre-activation; the residual and the old dense channels are never copied."""
        first = stage[0][1]
        c1c, inc, nb = first.c1c, first.inc, len(stage)
        if first.kind != "down" and not X.is_contiguous():
            X = X.contiguous()                                 # only bn_relu_s2 reads batch-strided taps
        if act is None:
            act = _bn_relu(first.c1x1_a.bn, X)
        if first.kind == "down":
            w = first.c1x1_w_s2
            pin = ops.bn_relu_s2(X, *_bn_consts(w.bn))
        else:
            w = first.c1x1_w_s1
            pin = _bn_relu(w.bn, X)
        N, Cin, Hs, Ws = pin.shape
        rows = c1c + 3 * inc + nb * inc
        buf = torch.empty((N, rows, Hs, Ws), device=X.device, dtype=torch.float32)
        buf3 = buf.view(N, rows, Hs * Ws)
        buf3[:, 2 * inc + c1c:3 * inc + c1c].zero_()
        ops.tuned_matmul(_proj_weight(w.conv, c1c), pin.view(N, Cin, Hs * Ws), out=buf3[:, :2 * inc + c1c])
        res_slot = buf3[:, 2 * inc:3 * inc + c1c]
        out = None
        prefix = (self.tap_prefix or {}).get(stage[-1][0], 0)
        if prefix:
            Co_last = c1c + 2 * inc + nb * inc
            whole = torch.empty((N, prefix + Co_last, Hs, Ws), device=X.device, dtype=torch.float32)
            out = whole[:, prefix:]
            out._dmn_base = whole          # (tensor._base is not available under inference_mode)
        for m, (name, block) in enumerate(stage, 1):
            y = block._y_path(act)
            wc = block.c1x1_c.conv.weight
            ops.tuned_matmul(wc.view(wc.shape[0], wc.shape[1]), y.view(N, y.shape[1], Hs * Ws), acc=res_slot)
            bn = stage[m][1].c1x1_a.bn if m < nb else next_bn
            act, tap = ops.dpn_stage_pass(buf, c1c, inc, c1c + 2 * inc + m * inc, _bn_consts(bn),
                                          want_out=(m == nb), out=out if m == nb else None)
        return tap, act

    def forward(self, x):
        if (self.fuse_glue and self.fusable and not self.training
                and not torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.float32):
            return self._forward_fused(x.contiguous())
        feats = []
        out = x
        for name, module in self.features.named_children():
            if name == "conv1_1":
                out, raw = module(out)
                feats.append(raw)
            else:
                out = module(out)
                if name in self.taps:
                    feats.append(_join(out))
        return out, feats

    def load_state_dict(self, new_state, strict=False):
        """Lenient by name, like dpn_model.py:376-381."""
        state = self.state_dict()
        for key in state:
            if key in new_state:
                state[key] = new_state[key]
        return super().load_state_dict(state)


def create_backbone(name, pretrained=False, extra=True):
    if name not in VARIANTS:
        raise NotImplementedError(
            "dms_amd: backbone %r is not available (only DPN variants %s; the torchvision "
            "backbones of the reference's model_factory are outside the DMN path)"
            % (name, sorted(VARIANTS)))
    if pretrained:
        raise RuntimeError(
            "dms_amd: pretrained=True would download ImageNet weights "
            "(dpn_model.py:37-54,103-104); construct with pretrained=False and load a "
            "state_dict instead")
    return DPN(name)
