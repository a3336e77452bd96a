# -*- coding: utf-8 -*-
"""oracle/dmn_ref.py -- TEST INFRASTRUCTURE ONLY (not product code).

Pure-torch CPU restatement of the reference's language-conditioned hot path, i.e.
everything ``BaseDMN.forward`` does AFTER the backbone (dmn_pytorch/models/dmn.py:136-215)
and ``UpsamplingModule.forward`` (dmn.py:328-341), written as functions over a flat
``{state_dict key: tensor}`` mapping that uses the reference's key names.

It deliberately keeps the reference's shape of computation -- one F.conv2d per word,
the materialised (L, C, H, W) concat, the full 1x1 comb_conv, the permute/contiguous,
the slice of the last H*W sequence steps -- so that it is an independent check of the
product's fused/split kernels.  The reference forward only works for batch 1 (the
``torch.cat`` at dmn.py:142 joins a batch-1 coordinate map); the semantics for B > 1
are "stack of B independent batch-1 forwards", which is what ``hot_path`` loops over.

Pinned against the reference's own dmn.py by tests/golden/*.npz (made by
oracle/gen_golden.py, which imports /root/reference with this package's SRURef bound
to the missing `sru` module).  The SRU arithmetic itself stays "parity unpinned"
(see oracle/sru_ref.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import numpy as np
import torch
import torch.nn.functional as F

from .sru_ref import sru_scan_torch


def spatial_batch(H, W):
    """8 coordinate channels, reference dmn.py:225-246 (float64 maths, stored f32)."""
    out = np.zeros((1, 8, H, W), dtype=np.float32)
    for h in range(H):
        for w in range(W):
            xmin = w / W * 2 - 1
            xmax = (w + 1) / W * 2 - 1
            xctr = (xmin + xmax) / 2
            ymin = h / H * 2 - 1
            ymax = (h + 1) / H * 2 - 1
            yctr = (ymin + ymax) / 2
            out[0, :, h, w] = [xmin, ymin, xmax, ymax, xctr, yctr, 1 / W, 1 / H]
    return torch.from_numpy(out)


def sru_stack(P, prefix, x, n_layers, act="tanh", scale_x=1.0, use_c=False):
    """`sru.SRU.forward` restated: per layer U = x.W then the scan; returns (h, c_last stack)."""
    prev, lst = x, []
    for i in range(n_layers):
        w = P["%s.rnn_lst.%d.weight" % (prefix, i)]
        b = P["%s.rnn_lst.%d.bias" % (prefix, i)]
        T, B, n_in = prev.shape
        d = b.numel() // 2
        k = w.shape[1] // d
        u = prev.contiguous().view(-1, n_in).mm(w).view(T, B, d, k)
        if use_c:
            from . import sru_ref
            hn, _, cl = sru_ref.sru_fwd_c(
                u.detach().numpy(), prev.detach().contiguous().numpy() if k == 3 else None,
                b.detach().numpy(), None, act, scale_x, want_c=False)
            h, c_last = torch.from_numpy(hn), torch.from_numpy(cl)
        else:
            h, c = sru_scan_torch(u, prev if k == 3 else None, b, None, act, scale_x)
            c_last = c[-1]
        prev = h
        lst.append(c_last)
    return prev, torch.stack(lst)


def hot_path_single(P, vis, lang, cfg, pre="langvis", trace=None, use_c=False):
    """One (image-features, query) pair.  vis (1, C, H, W) backbone output, lang (1, L) int64.

    cfg keys: num_filters, vis_size, lang_layers, mixed_layers, mix_we, high_res,
    activation, scale_x.  Mirrors dmn.py:136-215 line by line.
    """
    act, sx = cfg.get("activation", "tanh"), cfg.get("scale_x", 1.0)
    _, _, H, W = vis.shape
    vis = torch.cat([vis, spatial_batch(H, W)], dim=1)                  # dmn.py:138-142
    emb = F.embedding(lang, P[pre + ".emb.weight"]).transpose(0, 1)      # (L,1,E) dmn.py:147-149
    L = emb.shape[0]
    lin_in, mix = [], []
    if cfg["mix_we"]:
        lin_in.append(emb.squeeze(1))
    mix.append(emb[..., None, None].expand(L, 1, emb.shape[2], H, W))
    hl, _ = sru_stack(P, pre + ".lang_model", emb, cfg["lang_layers"], act, sx, use_c)
    mix.append(hl[..., None, None].expand(L, 1, hl.shape[2], H, W))
    if cfg["mix_we"]:
        lin_in.append(hl.squeeze(1))
        lin_in = torch.cat(lin_in, dim=1)                                # [WE, h] dmn.py:163-165
    else:
        lin_in = hl.squeeze(1)
    mix = torch.cat(mix, dim=2)                                          # dmn.py:170
    filt = torch.sigmoid(F.linear(lin_in, P[pre + ".adaptative_filter.weight"],
                                  P[pre + ".adaptative_filter.bias"]))    # dmn.py:173-174
    filt = filt.view(L, cfg["num_filters"], cfg["vis_size"] + 8, 1, 1)   # dmn.py:176-177
    p = torch.cat([F.conv2d(vis, filt[t]).unsqueeze(0) for t in range(L)])  # dmn.py:178-184
    q = torch.cat([vis.unsqueeze(0).expand(L, *vis.shape), mix, p], dim=2)  # dmn.py:187-189
    q = F.conv2d(q.squeeze(1), P[pre + ".comb_conv.weight"], P[pre + ".comb_conv.bias"])
    M = q.shape[1]
    seq = q.view(L, 1, M, H * W).permute(3, 0, 1, 2).contiguous().view(H * W * L, 1, M)
    out, _ = sru_stack(P, pre + ".mrnn", seq, cfg["mixed_layers"], act, sx, use_c)  # dmn.py:204
    if trace is not None:
        trace.update(filters=filt.view(L, cfg["num_filters"], -1), p=p.squeeze(1),
                     q=seq, lang_h=hl, mrnn_h=out)
    out = out[-(H * W):].permute(1, 2, 0).contiguous().view(1, -1, H, W)   # dmn.py:209-212
    if not cfg["high_res"]:
        out = F.conv2d(out, P[pre + ".output_collapse.weight"],
                       P[pre + ".output_collapse.bias"])                  # dmn.py:213-214
    return out


def hot_path(P, vis, lang, cfg, pre="langvis", use_c=False):
    """B >= 1: stack of independent batch-1 forwards (the reference is batch-1 only)."""
    return torch.cat([hot_path_single(P, vis[b:b + 1], lang[b:b + 1], cfg, pre, None, use_c)
                      for b in range(vis.shape[0])], dim=0)


def upsampling(P, x, features, n_levels, pre="upsample"):
    """UpsamplingModule.forward, dmn.py:328-341 (non_linearity=False, mode bilinear)."""
    i = len(features) - 1
    for lvl in range(n_levels):
        f = features[i]
        if x.shape[-2:] != f.shape[-2:]:
            x = F.interpolate(x, f.shape[-2:], mode="bilinear", align_corners=False)
        x = torch.cat([x, f], dim=1)
        x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)
        w = P["%s.convs.%d.1.weight" % (pre, lvl)]
        x = F.conv2d(x, w, P["%s.convs.%d.1.bias" % (pre, lvl)], padding=w.shape[-1] // 2)
        i -= 1
    return F.conv2d(x, P[pre + ".out_layer.weight"], P[pre + ".out_layer.bias"])
