# -*- coding: utf-8 -*-
"""Dynamic Multimodal Network, MI355X-native hot path behind the reference's module API.

Drop-in for ``dmn_pytorch.models.{DMN, BaseDMN}`` (reference dmn_pytorch/models/dmn.py):
same constructor arguments and defaults (dmn.py:87-91, 423-429), ``forward(vis, lang)``,
lenient ``load_state_dict`` (dmn.py:217-223, 457-463) and identical state_dict keys/shapes
-- but ``BaseDMN.forward`` (dmn.py:131-215) and ``UpsamplingModule.forward`` (dmn.py:328-341)
are re-designed around the HIP kernels of libdmn_hip.so:

    reference step (dmn.py)                       here
    ------------------------------------------   -------------------------------------------
    generate_spatial_batch + cat        138-142   ops.cat_coords  (device kernel, no H2D)
    SRU language encoder                    156   dms_amd.SRU     (GEMM + HIP scan)
    L x F.conv2d(vis, filters[t]) + cat 178-184   ops.dynfilter   (one batched thin GEMM)
    expand + cat (96 MB) + comb_conv    187-192   two small GEMMs + ops.mix_bcast_add
    view/permute/contiguous             196-201   fused into mix_bcast_add's store order
    mSRU over S*L steps                     204   dms_amd.SRU     (GEMM + HIP scan)
    output[-(H*W):] + permute           209-212   last scan stores only those S steps
    cat + Upsample(x2) per level        337-338   ops.cat_upsample2x (3x3 conv stays MIOpen)

Batches: the reference forward only works for B = 1 (its coordinate map has batch 1,
dmn.py:142); here B >= 1 and every sample is computed exactly as an independent batch-1
reference forward.  Tensors must live on a ROCm device: there is no CPU fallback.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

import logging

from .. import ops
from ..sru import SRU
from .dpn import create_backbone


# ops.mix_sru_fwd is bound by its waves' issue, not by HBM (~0.32-0.33 ms at T = 5120 however few columns:
# two cooperating waves per 64 columns, rank-10 update on the matrix cores, round 2), the mix + scan pair is
# HBM-bound and shrinks with the batch: measured cross-over on MI355X (tools/mix_sru_bench.py) between 5 and 6
# samples of 1005 columns (B=4: 0.26 vs 0.32 ms, B=8: 0.45 vs 0.32, B=16: 0.66 vs 0.32, B=32: 1.30 vs 0.33)
MIX_SCAN_MIN_COLS = 6000

_LOG = logging.getLogger("dms_amd")


def _lenient_load(module, new_state, verbose=True):
    """Copy same-name, same-size tensors; ignore the rest (dmn.py:217-223, 457-463).

    The reference skips silently.  Here every entry of the model that kept its current value
    (absent from `new_state`, or a different size) is named in ONE log line (logger "dms_amd",
    WARNING), together with checkpoint keys the model has no use for: the SRU parameter names
    (``rnn_lst.{i}.weight / .bias``) belong to the third-party `sru` package and are recalled,
    not verified, so a published ``.pth`` with other names would otherwise load as random SRU
    weights without a word.  BatchNorm ``num_batches_tracked`` counters (absent from
    PyTorch-0.4 checkpoints) are not reported.  Returns the list of skipped model keys."""
    state = module.state_dict()
    skipped = []
    for name in state:
        if name in new_state and state[name].size() == new_state[name].size():
            state[name] = new_state[name]
        else:
            skipped.append(name)
    nn.Module.load_state_dict(module, state)
    if verbose:
        report = [k for k in skipped if not k.endswith("num_batches_tracked")]
        unused = [k for k in new_state if k not in state]
        if report or unused:
            def why(k):
                return "%s (%s)" % (k, "absent" if k not in new_state else "size %s != %s" % (
                    tuple(new_state[k].size()), tuple(state[k].size())))
            _LOG.warning("dms_amd lenient load_state_dict: %d model entries kept their current values: %s; "
                         "%d checkpoint entries unused: %s", len(report), ", ".join(map(why, report)) or "-",
                         len(unused), ", ".join(unused) or "-")
    return skipped


class BaseDMN(nn.Module):
    def __init__(self, dict_size, emb_size=1000, hid_size=1000,
                 vis_size=2688, num_filters=10, mixed_size=1000,
                 hid_mixed_size=1005, lang_layers=3, mixed_layers=3,
                 backend='dpn92', mix_we=True, lstm=False, pretrained=True,
                 extra=True, high_res=False):
        super().__init__()
        self.high_res = high_res
        self.vis_size = vis_size
        self.num_filters = num_filters
        self.emb_size, self.hid_size = emb_size, hid_size
        self.mix_we = mix_we
        self.lstm = lstm
        # inference-time algebraic fold of comb_conv into the mSRU layer-0 projection
        # (see _folded_weights); set to False to run the unfused path
        self.fold_comb_into_msru = True
        # with the fold: mix fused into the layer-0 scan (ops.mix_sru_fwd) when the batch is large
        # enough for it to win (MIX_SCAN_MIN_COLS columns); False = mix kernel + scan kernel
        self.fuse_mix_scan = True
        # the fold under autograd too (round 2): W_fold is then built differentiably every step and
        # the gradients of comb_conv and of layer 0's weight arrive through it; layer 0's three
        # (S*L*B) x M x 4Hm GEMMs (forward, dgrad, wgrad) shrink to (S*B) x C x 4Hm ones
        self.fold_in_training = True
        # inference: the last mSRU scan stores its kept hidden states as the (B, Hm, H, W) map itself
        # (ops.sru_forward_image; dmn.py:209-212) instead of (S, B, Hm) + permute + contiguous
        self.scan_writes_image = True
        self._fold_cache = None
        self.base = create_backbone(backend, pretrained=pretrained, extra=extra)

        self.emb = nn.Embedding(dict_size, emb_size)
        if lstm:
            self.lang_model = nn.LSTM(emb_size, hid_size, num_layers=lang_layers)
        else:
            self.lang_model = SRU(emb_size, hid_size, num_layers=lang_layers)

        self.adaptative_filter = nn.Linear(
            in_features=hid_size + emb_size * int(mix_we),
            out_features=num_filters * (vis_size + 8))
        self.comb_conv = nn.Conv2d(
            in_channels=8 + emb_size + hid_size + vis_size + num_filters,
            out_channels=mixed_size, kernel_size=1, padding=0)
        if lstm:
            self.mrnn = nn.LSTM(mixed_size, hid_mixed_size, num_layers=mixed_layers)
        else:
            self.mrnn = SRU(mixed_size, hid_mixed_size, num_layers=mixed_layers)
        if not self.high_res:
            self.output_collapse = nn.Conv2d(in_channels=hid_mixed_size, out_channels=1,
                                             kernel_size=1)

    # -- the language-conditioned hot path, given backbone features -------------------
    def hot_path(self, vis, lang, image_dest=None):
        """vis (B, vis_size, H, W) backbone output, lang (B, L) int64 ->
        (B, hid_mixed, H, W) if high_res else (B, 1, H, W).
        image_dest (inference): a batch-strided (B, hid_mixed, H, W) view that the last mSRU scan
        writes the map into (the first channels of the head's [x | feat] buffer); the returned
        tensor is then that view."""
        with ops.gemm_backend():
            return self._hot_path(vis, lang, image_dest)

    def _image_arg(self, T, S, B, H, W, image_dest, layer0_external):
        """image_out argument of the mSRU call: None (the (S, B, Hm) form + permute), True (the scan
        writes a fresh (B, Hm, S) map) or the caller's destination as (B, Hm, S).  dmn.py:209-212."""
        if self.lstm or not self.scan_writes_image or not self.mrnn.can_image(T, S, layer0_external):
            return None
        if image_dest is not None:
            Hm = self.mrnn.n_out
            ok = (torch.is_tensor(image_dest) and image_dest.is_cuda and image_dest.dtype == torch.float32
                  and tuple(image_dest.shape) == (B, Hm, H, W) and image_dest[0].is_contiguous())
            if ok:
                return image_dest.flatten(2)
        return True

    def _hot_path(self, vis, lang, image_dest=None):
        B, C, H, W = vis.shape
        S = H * W
        C8 = C + 8
        E, Hd, K = self.emb_size, self.hid_size, self.num_filters

        V = ops.cat_coords(vis).view(B, C8, S)                    # dmn.py:138-142

        we = self.emb(lang.t())                                   # (L, B, E)  dmn.py:147-149 (gathered in that order)
        L = we.shape[0]
        hl, _ = self.lang_model(we)                               # (L, B, Hd) dmn.py:156
        lang_cat = torch.cat([we, hl], dim=2).view(L * B, E + Hd)  # [WE, h]
        lin_in = lang_cat if self.mix_we else hl.reshape(L * B, Hd)  # dmn.py:163-167

        grad = torch.is_grad_enabled()
        if grad:
            filt = torch.addmm(self.adaptative_filter.bias, lin_in,
                               self.adaptative_filter.weight.t()).sigmoid_()   # dmn.py:173-174
        else:                                                      # bias-free GEMM, then bias + sigmoid in one pass
            filt = ops.bias_sigmoid_(torch.mm(lin_in, self.adaptative_filter.weight.t()),
                                     self.adaptative_filter.bias)
        filt = filt.view(L, B, K, C8)                              # dmn.py:176-177
        P = ops.dynfilter(V, filt)                                 # (L,B,K,S) dmn.py:178-184

        fold = (self.fold_comb_into_msru and not self.lstm and self.mrnn.rnn_lst[0].k == 4
                and (not grad or (self.fold_in_training and K <= 10 and L <= 24)))
        if fold:
            Wv_f, Wl_f, Wp_f, b_f = self._folded_weights()
            if grad:                                                # autograd-visible library GEMMs
                A = torch.matmul(V.transpose(1, 2), Wv_f)
            else:
                A = ops.tuned_matmul(V.transpose(1, 2), Wv_f)       # (B, S, 4*Hm)
            Bt = torch.addmm(b_f, lang_cat, Wl_f).view(L, B, -1)    # (L, B, 4*Hm)
            cell0 = self.mrnn.rnn_lst[0]
            if (not grad and self.fuse_mix_scan and K <= 16 and B * cell0.n_out >= MIX_SCAN_MIN_COLS):
                # enough columns to occupy the SIMDs: the scan lanes build U0 themselves
                img = self._image_arg(S * L, S, B, H, W, image_dest, True)
                h0, c0_last = ops.mix_sru_fwd(A, Bt, P, Wp_f, cell0.bias, None, cell0.activation)
                out, _ = self.mrnn(h0, tail=S, first_done=c0_last, image_out=img)
            else:
                img = self._image_arg(S * L, S, B, H, W, image_dest, True) if not grad else None
                U0 = ops.mix_bcast_add(A, Bt, P, Wp_f)              # (S*L, B, 4*Hm) = layer-0 U
                out, _ = self.mrnn(U0, tail=S, projected=True, image_out=img)
            if img is None:
                out = out.permute(1, 2, 0).contiguous()             # dmn.py:210-212
            out = out.view(B, -1, H, W)
            if not self.high_res:
                oc = self.output_collapse                              # dmn.py:213-214
                if grad:
                    out = oc(out)
                else:
                    out = ops.channel_dot(out, oc.weight.detach().view(-1), oc.bias.detach())
            return out

        # comb_conv split by input block [vis+coords | WE | h | P]  (dmn.py:114-118,187-192)
        Wc = self.comb_conv.weight.view(self.comb_conv.out_channels, -1)
        Wv, Wl, Wp = Wc[:, :C8], Wc[:, C8:C8 + E + Hd], Wc[:, C8 + E + Hd:]
        A = torch.matmul(V.transpose(1, 2), Wv.t())                # (B, S, M)
        Bt = torch.addmm(self.comb_conv.bias, lang_cat, Wl.t()).view(L, B, -1)
        Q = ops.mix_bcast_add(A, Bt, P, Wp)                        # (S*L, B, M) dmn.py:192-201

        if self.lstm:
            out, _ = self.mrnn(Q)
            out = out[-S:]
            out = out.permute(1, 2, 0).contiguous()
        else:
            img = self._image_arg(S * L, S, B, H, W, image_dest, False) if not grad else None
            out, _ = self.mrnn(Q, tail=S, image_out=img)           # (S, B, Hm) dmn.py:204-209
            if img is None:
                out = out.permute(1, 2, 0).contiguous()            # dmn.py:210-212
        out = out.view(B, -1, H, W)
        if not self.high_res:
            out = self.output_collapse(out)                        # dmn.py:213-214
        return out

    def _folded_weights(self):
        """comb_conv (dmn.py:192) followed by the mSRU layer-0 projection x.W0 (dmn.py:204) is
        one linear map: W_fold = Wc^T . W0  (4706 x 4*Hm), b_fold = bc . W0.  With it
        U0 = q.W_fold + b_fold is assembled directly from the vis / language / P blocks and the
        (S*L, B, M) tensor Q and its 20.6 GMAC/sample GEMM are never needed (3.9 GMAC instead).
        Exact algebra, fp32 re-association only (~1e-6 relative; parity-tested).  Cached per
        weight version at inference; under autograd it is rebuilt as part of the graph."""
        Wc = self.comb_conv.weight
        W0 = self.mrnn.rnn_lst[0].weight
        n_lang = self.emb_size + self.hid_size
        if torch.is_grad_enabled():                                  # training: part of the graph, no cache
            Wfold = Wc.view(Wc.shape[0], -1).t().mm(W0)
            bfold = self.comb_conv.bias.view(1, -1).mm(W0).view(-1)
            c8 = Wfold.shape[0] - n_lang - self.num_filters
            return (Wfold[:c8], Wfold[c8:c8 + n_lang], Wfold[c8 + n_lang:].t().contiguous(), bfold)
        key = (Wc.data_ptr(), Wc._version, W0.data_ptr(), W0._version,
               self.comb_conv.bias._version, str(Wc.device))
        if self._fold_cache is None or self._fold_cache[0] != key:
            with torch.no_grad():
                Wfold = Wc.view(Wc.shape[0], -1).t().mm(W0)               # (4706, 4*Hm)
                bfold = self.comb_conv.bias.view(1, -1).mm(W0).view(-1)   # (4*Hm,)
                n_lang = self.emb_size + self.hid_size
                c8 = Wfold.shape[0] - n_lang - self.num_filters
                parts = (Wfold[:c8].contiguous(), Wfold[c8:c8 + n_lang].contiguous(),
                         Wfold[c8 + n_lang:].t().contiguous(), bfold)     # Wp' as (4*Hm, K)
            self._fold_cache = (key, parts)
        return self._fold_cache[1]

    def forward(self, vis, lang):
        vis, base_features = self.base(vis)                        # dmn.py:134
        # high-res inference: the backbone left room for the map in front of its last pyramid tap
        # (DMN._link_head_buffers); the last mSRU scan writes it there, cat([x, feat]) costs nothing
        dest = None
        if self.high_res and not torch.is_grad_enabled() and base_features:
            whole = getattr(base_features[-1], "_dmn_base", None)
            Hm = self.mrnn.n_out if not self.lstm else -1
            if (torch.is_tensor(whole) and whole.dim() == 4 and whole.is_contiguous()
                    and whole.shape[1] == Hm + base_features[-1].shape[1]
                    and tuple(whole.shape[2:]) == tuple(vis.shape[2:]) and whole.shape[0] == vis.shape[0]):
                dest = whole[:, :Hm]
        return self.hot_path(vis, lang, image_dest=dest), base_features

    def load_state_dict(self, new_state, strict=False):
        return _lenient_load(self, new_state)

    @classmethod
    def generate_spatial_batch(cls, featmap_H, featmap_W, device=None):
        """(1, 8, H, W) coordinate map computed on the device (dmn.py:225-246)."""
        device = device or torch.device("cuda")
        z = torch.zeros(1, 0, featmap_H, featmap_W, device=device)
        return ops.cat_coords(z)


WINO_MIN_COUT = 64      # "auto": narrowest level that still goes through the Winograd GEMMs


class _Up(nn.Module):
    """Parameter-free placeholder at index 0 of each level so that the convolution keeps
    the reference's key ``convs.{i}.1`` (dmn.py:322-324)."""

    def __init__(self, mode):
        super().__init__()
        self.mode = mode

    def forward(self, x):
        return F.interpolate(x, scale_factor=2, mode=self.mode,
                             align_corners=False if self.mode == 'bilinear' else None)


class UpsamplingModule(nn.Module):
    def __init__(self, in_channels, mode='bilinear', ker_size=3,
                 amplification=32, non_linearity=False,
                 feature_channels=[2688, 1552, 704, 336, 64]):
        super().__init__()
        self.ker_size = ker_size
        self.non_linearity = non_linearity
        self.mode = mode
        self.up = _Up(mode)
        num_layers = int(np.log2(amplification))
        levels = []
        for i, out_channels in enumerate(np.logspace(
                9, 10 - num_layers, num=num_layers, base=2, dtype=int)):
            conv = nn.Conv2d(int(in_channels) + feature_channels[i], int(out_channels),
                             kernel_size=ker_size, padding=ker_size // 2)
            seq = [self.up, conv] + ([nn.PReLU()] if non_linearity else [])
            levels.append(nn.Sequential(*seq))
            in_channels = int(out_channels)
        self.out_layer = nn.Conv2d(in_channels, 1, kernel_size=1, padding=0)
        self.convs = nn.ModuleList(levels)
        # inference: run a level's cat + upsample + 3x3 conv as fused Winograd transforms around a
        # batched rocBLAS GEMM.  "auto" = F(4x4) where measured faster than MIOpen, True / "f4" =
        # every eligible level, "f2" = F(2x2) (error ~7e-7 instead of ~1e-5 per level), False = never.
        self.winograd = "auto"
        self.fuse_last = True      # inference: last level + out_layer as one C -> 1 conv
        # inference: a level's 3x3 conv applied as nine tap matrices at LOW resolution (one plain
        # library GEMM; upsampling and channel mixing commute) followed by one gather pass at
        # high resolution (ops.tap_gemm / tap_upsample_gather).  Same multiply count as F(4x4)
        # Winograd, exact fp32 products, no transformed-input tensor.  Takes precedence over
        # `winograd`; False falls back to it.
        self.tap_gemm = True

    def _tap_ok(self, conv, x):
        return (self.tap_gemm and self.mode == 'bilinear' and not torch.is_grad_enabled()
                and x.is_cuda and x.dtype == torch.float32
                and conv.kernel_size == (3, 3) and conv.padding == (1, 1) and conv.stride == (1, 1)
                and conv.dilation == (1, 1) and conv.groups == 1)

    @staticmethod
    def _concat_base_for(f, B, Cx, H, W):
        """The (B, Cx + Cf, H, W) buffer whose last Cf channels are the view `f`, or None."""
        base = getattr(f, "_dmn_base", None)
        if base is None or not torch.is_tensor(base) or base.dim() != 4 or not base.is_contiguous():
            return None
        if f.dim() != 4 or tuple(f.shape[2:]) != (H, W) or f.shape[0] != B:
            return None
        Cf = f.shape[1]
        if tuple(base.shape) != (B, Cx + Cf, H, W) or f.stride() != (base.stride(0), H * W, W, 1):
            return None
        if f.storage_offset() != base.storage_offset() + Cx * H * W or not f.is_cuda or f.dtype != torch.float32:
            return None
        return base

    def _concat_base(self, x, f):
        return self._concat_base_for(f, x.shape[0], x.shape[1], x.shape[2], x.shape[3])

    def _tap_weight_full(self, conv):
        """Wt (9*Cout, Cx+Cf) of a level's conv weight in tap-major row order, cached per weight version."""
        w = conv.weight
        key = (w._version, w.data_ptr())
        cached = conv.__dict__.get("_dmn_tap_full")
        if cached is None or cached[0] != key:
            with torch.no_grad():
                cached = (key, ops.tap_weight(w.detach()))
            conv.__dict__["_dmn_tap_full"] = cached
        return cached[1]

    def _tap_weights(self, conv, cx):
        """(Wt[:, :cx], Wt[:, cx:]) of a level's conv weight in tap-major row order, cached per
        weight version."""
        w = conv.weight
        key = (w._version, w.data_ptr(), cx)
        cached = conv.__dict__.get("_dmn_tap")
        if cached is None or cached[0] != key:
            with torch.no_grad():
                wt = ops.tap_weight(w.detach())
                cached = (key, (wt[:, :cx].contiguous(), wt[:, cx:].contiguous()))
            conv.__dict__["_dmn_tap"] = cached
        return cached[1]

    def _wino_weight(self, conv, m):
        """U = G g G^T of a level's conv weight, cached per weight version and tile size."""
        w = conv.weight
        key = (w._version, w.data_ptr(), m)
        cached = conv.__dict__.get("_dmn_wino")
        if cached is None or cached[0] != key:
            with torch.no_grad():
                cached = (key, ops.wino_filter_transform(w.detach(), m))
            conv.__dict__["_dmn_wino"] = cached
        return cached[1]

    def _winograd_tile(self, conv, x):
        """0 = MIOpen path, else the Winograd output tile size m for this level."""
        mode = self.winograd
        if (mode is False or self.mode != 'bilinear' or torch.is_grad_enabled()
                or not x.is_cuda or conv.kernel_size != (3, 3) or conv.padding != (1, 1)
                or conv.stride != (1, 1) or conv.dilation != (1, 1) or conv.groups != 1):
            return 0
        even = x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0
        if mode == "f2":
            return 2
        if mode is True or mode == "f4":
            return 4 if even else 2
        # "auto", measured on MI355X (profiles/r1_wino_probe.txt): the batched GEMMs win while
        # the level is wide; narrow levels turn memory-bound on V and MIOpen's kernels are faster
        if conv.out_channels >= WINO_MIN_COUT and conv.in_channels >= 256:
            return 4 if even else 2
        return 0

    def forward(self, x, features):
        i = len(features) - 1
        for level in self.convs:
            f = features[i]
            if x.shape[-2:] != f.shape[-2:]:                      # dmn.py:332-336
                x = F.interpolate(x, f.shape[-2:], mode='bilinear', align_corners=False)
            conv = level[1]
            if level is self.convs[-1] and self._can_fuse_last(level, conv, x):
                weff, b_eff, w9 = self._final_weights(conv)
                if self._tap_ok(conv, x):
                    return ops.tap_upsample_gather(ops.tap_project(x.contiguous(), f.contiguous(), w9), b_eff)
                return ops.head_final(x.contiguous(), f.contiguous(), weff, b_eff)
            m = 0 if self._tap_ok(conv, x) else self._winograd_tile(conv, x)
            if self._tap_ok(conv, x):
                rest = list(level)[2:]
                base = self._concat_base(x, f)
                if base is not None:
                    # the backbone wrote this tap behind room for x: [x | feat] is ONE tensor and the
                    # level ONE GEMM with K = Cx + Cf (no second GEMM re-reading Z with beta = 1)
                    B, Cx, H, W = x.shape
                    if x.data_ptr() != base.data_ptr():
                        base[:, :Cx].copy_(x)
                    Z = ops.tuned_matmul(self._tap_weight_full(conv), base.view(B, base.shape[1], H * W))
                    Z = Z.view(B, 9, conv.out_channels, H, W)
                else:
                    wt_x, wt_f = self._tap_weights(conv, x.shape[1])
                    Z = ops.tap_gemm(x.contiguous(), f.contiguous(), wt_x, wt_f)
                # write straight into the next level's [x | feat] buffer when there is one
                dest = None
                if not rest and i >= 1 and level is not self.convs[-1]:
                    nb_ = self._concat_base_for(features[i - 1], x.shape[0], conv.out_channels,
                                                2 * x.shape[2], 2 * x.shape[3])
                    if nb_ is not None:
                        dest = nb_[:, :conv.out_channels]
                x = ops.tap_upsample_gather(Z, conv.bias, out=dest)
            elif m:
                B, _, H, W = x.shape
                V = ops.wino_cat_upsample2x_transform(x.contiguous(), f.contiguous(), m)
                M = ops.tuned_matmul(self._wino_weight(conv, m), V)
                x = ops.wino_output_transform(M, conv.bias, B, H, W, m)
                rest = list(level)[2:]
            else:
                if self.mode == 'bilinear':
                    x = ops.cat_upsample2x(x, f)                  # dmn.py:337 + Upsample
                else:
                    x = self.up(torch.cat([x, f], dim=1))
                rest = list(level)[1:]
            last = level is self.convs[-1]
            if (last and rest and rest[0] is conv and len(rest) == 1 and conv.bias is not None
                    and self._can_reduce(x)):
                # last level on MIOpen: its bias folds into out_layer's (both are linear), which
                # saves the separate bias pass over the 512^2 x 32-channel tensor
                x = F.conv2d(x, conv.weight, None, conv.stride, conv.padding)
                return ops.channel_dot(x, self.out_layer.weight.view(-1), self._out_bias(conv))
            for layer in rest:
                x = layer(x)
            i -= 1
        if self._can_reduce(x):
            return ops.channel_dot(x, self.out_layer.weight.view(-1), self._out_bias(None))
        return self.out_layer(x)

    def _can_fuse_last(self, level, conv, x):
        """Last level + out_layer as one single-output-channel 3x3 conv (no non-linearity between
        them): inference, bilinear, plain 3x3 / pad 1 conv, out_layer to one channel."""
        return (self.fuse_last and len(level) == 2 and self.mode == 'bilinear' and self._can_reduce(x)
                and conv.kernel_size == (3, 3) and conv.padding == (1, 1) and conv.stride == (1, 1)
                and conv.dilation == (1, 1) and conv.groups == 1
                and self.out_layer.kernel_size == (1, 1))

    def _final_weights(self, conv):
        """w_eff (C, 3, 3) = sum_o w_out[o] * w[o], b_eff (1) = b_out + sum_o w_out[o] * b[o] and
        w_eff again as (9, C) tap rows; cached per parameter version."""
        ol = self.out_layer
        ts = [conv.weight, ol.weight, ol.bias] + ([conv.bias] if conv.bias is not None else [])
        key = tuple((t._version, t.data_ptr()) for t in ts)
        cached = self.__dict__.get("_dmn_final_w")
        if cached is None or cached[0] != key:
            with torch.no_grad():
                wo = ol.weight.detach().view(-1).double()
                weff = torch.einsum("o,ocyx->cyx", wo, conv.weight.detach().double())
                b = ol.bias.detach().double().view(1)
                if conv.bias is not None:
                    b = b + (wo * conv.bias.detach().double()).sum().view(1)
            weff = weff.float().contiguous()
            cached = (key, (weff, b.float().contiguous(), weff.view(weff.shape[0], 9).t().contiguous()))
            self.__dict__["_dmn_final_w"] = cached
        return cached[1]

    def _can_reduce(self, x):
        return (not torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.float32
                and self.out_layer.out_channels == 1)

    def _out_bias(self, conv):
        """out_layer.bias (+ out_layer.weight . conv.bias when the preceding conv ran bias-free),
        as a 1-element device tensor; cached per parameter version."""
        ol = self.out_layer
        ts = [ol.weight, ol.bias] + ([conv.bias] if conv is not None else [])
        key = tuple((t._version, t.data_ptr()) for t in ts)
        cached = self.__dict__.get("_dmn_out_bias")
        if cached is None or cached[0] != key:
            with torch.no_grad():
                b = ol.bias.detach().clone().view(1)
                if conv is not None:
                    b = b + (ol.weight.detach().view(-1) * conv.bias.detach()).sum().view(1)
            cached = (key, b.contiguous())
            self.__dict__["_dmn_out_bias"] = cached
        return cached[1]


class DMN(nn.Module):
    def __init__(self, dict_size, emb_size=1000, hid_size=1000,
                 vis_size=2688, num_filters=1, mixed_size=1000,
                 hid_mixed_size=1005, lang_layers=2, mixed_layers=3,
                 backend='dpn92', mix_we=False, lstm=False, pretrained=True,
                 extra=True, high_res=False,
                 upsampling_mode='bilineal', upsampling_size=3,
                 upsampling_amplification=32, dmn_freeze=False):
        super().__init__()
        self.langvis = BaseDMN(dict_size, emb_size, hid_size,
                               vis_size, num_filters, mixed_size,
                               hid_mixed_size, lang_layers, mixed_layers,
                               backend, mix_we, lstm, pretrained,
                               extra, high_res)
        self.high_res = high_res
        self.dmn_freeze = dmn_freeze
        if high_res:
            if upsampling_mode not in ('bilinear', 'nearest'):
                # the reference's default 'bilineal' (sic, dmn.py:428) only fails at the
                # first forward inside nn.Upsample; fail at construction instead
                raise ValueError("upsampling_mode must be 'bilinear' or 'nearest', got %r "
                                 "(the reference's default 'bilineal' is a typo that "
                                 "raises at its first forward)" % (upsampling_mode,))
            self.upsample = UpsamplingModule(
                hid_mixed_size, mode=upsampling_mode, ker_size=upsampling_size,
                amplification=upsampling_amplification)
            self._link_head_buffers(hid_mixed_size)
        if dmn_freeze:
            self.langvis.eval()

    def _link_head_buffers(self, hid_mixed_size):
        """Inference: let the backbone write each pyramid tap behind room for the head level's own
        input channels (DPN.tap_prefix), so that cat([x, feat]) (dmn.py:337) exists without a copy:
        level i reads features[4 - i]; its x has hid_mixed_size (i = 0) or the previous level's
        out_channels."""
        base = self.langvis.base
        taps = getattr(base, "taps", None)
        if not taps or len(taps) != 4 or not hasattr(base, "tap_prefix"):
            return
        prefix, cx = {}, int(hid_mixed_size)
        for i, level in enumerate(self.upsample.convs):
            k = 4 - i                                           # index into features
            if k >= 1:
                prefix[taps[k - 1]] = cx
            cx = level[1].out_channels
        base.tap_prefix = prefix

    def forward(self, vis, lang):
        if self.dmn_freeze:
            vis, lang = vis.detach(), lang.detach()
        out, features = self.langvis(vis, lang)
        if self.dmn_freeze:
            out = out.detach()
        if self.high_res:
            out = self.upsample(out, features)
        return out

    def load_state_dict(self, new_state, strict=False):
        return _lenient_load(self, new_state)
