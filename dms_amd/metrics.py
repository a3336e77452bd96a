# -*- coding: utf-8 -*-
"""Device-resident segmentation metrics (the reference's ``evaluate()``, train.py:341-434).

``threshold_iou`` is the HIP threshold sweep; ``SegmentationEvaluator`` keeps the
reference's accumulators -- cumulative I / U per threshold (train.py:345-346,376-377),
precision@{.5,.6,.7,.8,.9} counts (train.py:347-349,380-382) -- on the device, so a whole
validation run needs no host sync until ``compute()``.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import _lib
from .ops import _chk, _ptr, _stream

# train.py:343-344
SCORE_THRESH = np.concatenate([np.arange(start=0.00, stop=0.96, step=0.025)]).tolist()
EVAL_SEG_IOU = [.5, .6, .7, .8, .9]


def threshold_iou(prob, mask, thresholds):
    """prob, mask (B, ...) float32 cuda; thresholds (NT,) ascending float32 cuda.
    -> inter (B, NT), union (B, NT) as compute_mask_IU (train.py:333-338) would give them."""
    B = prob.shape[0]
    prob = prob.reshape(B, -1).contiguous()
    mask = mask.reshape(B, -1).contiguous()
    if prob.shape != mask.shape:
        raise RuntimeError("dms_amd.threshold_iou: prob %s vs mask %s" % (tuple(prob.shape), tuple(mask.shape)))
    _chk("threshold_iou", prob)
    _chk("threshold_iou", mask)
    _chk("threshold_iou", thresholds)
    NT = thresholds.numel()
    with torch.cuda.device(prob.device):
        inter = torch.zeros((B, NT), device=prob.device, dtype=torch.float32)
        union = torch.zeros((B, NT), device=prob.device, dtype=torch.float32)
        _lib.call("dmn_threshold_iou", _ptr(prob), _ptr(mask), _ptr(thresholds), _ptr(inter),
                  _ptr(union), B, prob.shape[1], NT, _stream())
    return inter, union


class SegmentationEvaluator(object):
    def __init__(self, device, score_thresh=None, seg_iou=None):
        st = list(SCORE_THRESH if score_thresh is None else score_thresh)
        if any(b <= a for a, b in zip(st, st[1:])):
            raise ValueError("thresholds must be strictly ascending")
        self.score_thresh = st
        self.seg_iou = list(EVAL_SEG_IOU if seg_iou is None else seg_iou)
        self.device = torch.device(device)
        # python floats -> float32, which is what `out > thresh` compares against (train.py:369)
        self.th = torch.tensor(st, dtype=torch.float32, device=self.device)
        self.iou_levels = torch.tensor(self.seg_iou, dtype=torch.float32, device=self.device)
        self.reset()

    def reset(self):
        nt = len(self.score_thresh)
        self.cum_I = torch.zeros(nt, device=self.device)
        self.cum_U = torch.zeros(nt, device=self.device)
        self.seg_correct = torch.zeros(len(self.seg_iou), nt, device=self.device)
        self.seg_total = 0

    @torch.no_grad()
    def update(self, logits, mask):
        """logits (B,1,h,w) network output; mask (B,H,W) or (B,1,H,W) target.  Follows
        train.py:360-382: sigmoid -> bilinear upsample to the mask size -> threshold sweep."""
        B = logits.shape[0]
        mask = mask.reshape(B, 1, mask.shape[-2], mask.shape[-1]).float()
        out = torch.sigmoid(logits)
        out = F.interpolate(out, size=mask.shape[-2:], mode='bilinear', align_corners=False)
        inter, union = threshold_iou(out, mask, self.th)
        self.cum_I += inter.sum(0)
        self.cum_U += union.sum(0)
        this_iou = inter / union                                   # (B, NT); 0/0 -> nan -> False
        self.seg_correct += (this_iou[:, None, :] >= self.iou_levels[None, :, None]).float().sum(0)
        self.seg_total += B
        return inter, union

    def compute(self):
        """One host sync: final IoUs per threshold, the maximising threshold, precision@X."""
        ious = (self.cum_I / self.cum_U).cpu()
        max_iou, max_idx = torch.max(ious, 0)
        prec = (self.seg_correct / max(self.seg_total, 1)).cpu()
        return {"ious": ious.tolist(), "max_iou": float(max_iou), "max_idx": int(max_idx),
                "threshold": self.score_thresh[int(max_idx)], "precision": prec.tolist(),
                "seg_total": self.seg_total}
