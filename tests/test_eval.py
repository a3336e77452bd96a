"""Evaluation metrics: oracle vs the reference's compute_mask_IU golden (CPU) and the HIP
threshold sweep vs the oracle (GPU)."""
import numpy as np
import pytest
import torch

from oracle import eval_ref
from oracle.detgen import det_uniform
from tests.util import load_golden


def _golden_inputs():
    g = load_golden("eval_iou")
    N, H, W = int(g["N"]), int(g["H"]), int(g["W"])
    prob = torch.sigmoid(det_uniform((N, H, W), 501, 4.0))
    prob[0, 0, :5] = torch.tensor(eval_ref.SCORE_THRESH[10:15], dtype=torch.float32)   # exact ties
    mask = (det_uniform((N, H, W), 502, 1.0) > 0.2).float()
    mask[2] = 0
    return g, prob, mask


def test_oracle_sweep_bit_exact_vs_reference_golden():
    g, prob, mask = _golden_inputs()
    assert np.allclose(g["thresholds"], eval_ref.SCORE_THRESH) and len(eval_ref.SCORE_THRESH) == 39
    for n in range(prob.shape[0]):
        i, u = eval_ref.sweep(prob[n], mask[n])
        np.testing.assert_array_equal(i.numpy(), g["inter"][n])
        np.testing.assert_array_equal(u.numpy(), g["union"][n])


@pytest.mark.gpu
def test_hip_threshold_sweep_bit_exact_vs_reference_golden():
    from dms_amd.metrics import threshold_iou
    g, prob, mask = _golden_inputs()
    th = torch.tensor(eval_ref.SCORE_THRESH, dtype=torch.float32, device="cuda")
    inter, union = threshold_iou(prob.cuda(), mask.cuda(), th)
    np.testing.assert_array_equal(inter.cpu().numpy(), g["inter"])
    np.testing.assert_array_equal(union.cpu().numpy(), g["union"])


@pytest.mark.gpu
def test_evaluator_matches_oracle_evaluate_full_size():
    from dms_amd.metrics import SegmentationEvaluator
    B = 5
    logits = det_uniform((B, 1, 16, 16), 601, 3.0)
    masks = (det_uniform((B, 480, 512), 602, 1.0) > 0.3).float()
    masks[1] = 0                                                   # union can be 0 -> nan IoU
    ev = SegmentationEvaluator("cuda")
    ev.update(logits[:2].cuda(), masks[:2].cuda())
    ev.update(logits[2:].cuda(), masks[2:].cuda())
    got = ev.compute()
    # bilinear upsampling differs in the last ulp between CPU and GPU; pixels that sit
    # within 1e-6 of a threshold may flip, so compare counts with a tiny tolerance
    ref = eval_ref.evaluate([logits[b:b + 1] for b in range(B)], [masks[b] for b in range(B)])
    np.testing.assert_allclose(ev.cum_I.cpu().numpy(), ref["cum_I"].numpy(), rtol=2e-4, atol=8)
    np.testing.assert_allclose(ev.cum_U.cpu().numpy(), ref["cum_U"].numpy(), rtol=2e-4, atol=8)
    assert abs(got["max_iou"] - ref["max_iou"]) < 1e-4 and got["seg_total"] == B
    assert np.abs(np.array(got["precision"]) - ref["precision"].numpy()).max() <= 1.0 / B + 1e-6
