"""Evaluation helpers either side of the hot path: the reference's end-point-error /
outlier metrics and its disparity rescale, so synthetic parity turns into benchmark EPE once a dataset
is mounted.  Reference: ``losses/squence_loss.py:6-26``, ``utils/metric.py:5-32``,
``dataloader/preprocess.py:257-272``.  Metrics are tiny reductions (host-framework ops on whatever
device the tensors live on); ``scale_disp`` uses the HIP resize kernel."""
from __future__ import annotations

import torch

from .upsample import resize_bilinear


def EPE_Loss(disp_infer, disp_gt, max_disp=192):
    """Mean |pred - gt| over pixels with 0 < gt < 192 (``squence_loss.py:22-26``)."""
    mask = (disp_gt > 0) & (disp_gt < max_disp)
    return (disp_infer[mask] - disp_gt[mask]).abs().mean()


def sequence_loss(pred_disp_list, gt_disp, gamma=0.8, max_dispariy=192):
    """Exponentially weighted masked L1 over the iteration outputs (``squence_loss.py:6-18``); note the
    reference averages over ALL pixels, masked ones contributing zero."""
    n = len(pred_disp_list)
    mask = (gt_disp < 192).float() * (gt_disp > 0).float()
    total = 0.0
    for i in range(n):
        w = gamma ** (n - i - 1)
        total = total + (w * (pred_disp_list[i] * mask - gt_disp * mask).abs()).mean()
    return total


def D1_metric(D_pred, D_gt):
    """KITTI D1: error > 3 px and > 5 % (``utils/metric.py:5-8``)."""
    E = torch.abs(D_pred - D_gt)
    return torch.mean(((E > 3) & (E / D_gt.abs() > 0.05)).float())


def P1_metric(D_pred, D_gt):
    """Fraction of pixels with error > 1 px (``utils/metric.py:10-13``)."""
    return torch.mean((torch.abs(D_pred - D_gt) > 1).float())


def thres_metric(d_est, d_gt, mask, thres):
    """Fraction of masked pixels with error > thres (``utils/metric.py:16-32``)."""
    assert isinstance(thres, (int, float))
    return torch.mean((torch.abs(d_gt[mask] - d_est[mask]) > thres).float())


def scale_disp(disp, output_size=(1, 540, 960)):
    """Bilinear resize of a [N,1,H,W] disparity to ``output_size[-2:]`` and rescale by the width ratio
    (``dataloader/preprocess.py:257-272``; nn.Upsample(bilinear) == align_corners=False)."""
    i_w = disp.size()[-1]
    o_w = output_size[-1]
    return resize_bilinear(disp, (output_size[-2], output_size[-1]), align_corners=False, scale=o_w * 1.0 / i_w)
