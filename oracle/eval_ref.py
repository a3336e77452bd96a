# -*- coding: utf-8 -*-
"""oracle/eval_ref.py -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's evaluation arithmetic (dmn_pytorch/train.py):
``compute_mask_IU`` (:333-338) and the accumulation of ``evaluate()`` (:343-349, 360-382,
419-421).  Pinned by tests/golden/eval_iou.npz (the reference's own compute_mask_IU run by
oracle/gen_golden.py)."""
import numpy as np
import torch
import torch.nn.functional as F

SCORE_THRESH = np.concatenate([np.arange(start=0.00, stop=0.96, step=0.025)]).tolist()
EVAL_SEG_IOU = [.5, .6, .7, .8, .9]


def mask_iu(masks, target):
    temp = masks * target
    return temp.sum(), ((masks + target) - temp).sum()


def sweep(prob, mask, score_thresh=SCORE_THRESH):
    """prob, mask (H, W) float32 -> inter, union (NT,) float32 (train.py:366-374)."""
    inter = torch.zeros(len(score_thresh))
    union = torch.zeros(len(score_thresh))
    for idx, thresh in enumerate(score_thresh):
        inter[idx], union[idx] = mask_iu((prob > thresh).float(), mask)
    return inter, union


def evaluate(logits_list, mask_list, score_thresh=SCORE_THRESH, seg_iou=EVAL_SEG_IOU):
    cum_I = torch.zeros(len(score_thresh))
    cum_U = torch.zeros(len(score_thresh))
    seg_correct = torch.zeros(len(seg_iou), len(score_thresh))
    total = 0
    for logits, mask in zip(logits_list, mask_list):
        out = torch.sigmoid(logits)
        out = F.interpolate(out, size=mask.shape[-2:], mode='bilinear', align_corners=False).squeeze()
        inter, union = sweep(out, mask)
        cum_I += inter
        cum_U += union
        this_iou = inter / union
        for i, lvl in enumerate(seg_iou):
            for j in range(len(score_thresh)):
                seg_correct[i, j] += float(this_iou[j] >= lvl)
        total += 1
    ious = cum_I / cum_U
    max_iou, max_idx = torch.max(ious, 0)
    return {"ious": ious, "max_iou": float(max_iou), "max_idx": int(max_idx),
            "precision": seg_correct / max(total, 1), "cum_I": cum_I, "cum_U": cum_U}
