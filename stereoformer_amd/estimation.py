"""Soft-argmin disparity regression (A4) behind the reference's surfaces:
``models/utils/estimation.py:7-62`` and ``models/utils/submodule.py:23-27``."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _native as N


def _soft_argmin(cost_volume, match_similarity, with_prob, who):
    assert cost_volume.dim() == 4  # estimation.py:15
    vol = N.dev_f32(cost_volume, f"{who}: cost_volume")
    B, D, H, W = vol.shape
    disp = torch.empty((B, H, W), device=vol.device, dtype=torch.float32)
    prob = torch.empty_like(vol) if with_prob else None
    with torch.cuda.device(vol.device):
        N.call("stereo_soft_argmin_f32", N.ptr(vol), N.ptr(disp), N.ptr(prob), B, D, H, W,
               1 if match_similarity else 0, N.stream_ptr(vol.device))
    return disp, prob


class DisparityEstimation(nn.Module):
    """``DisparityEstimation(max_disp, match_similarity=True)`` (estimation.py:7-33).

    Candidates are ``arange(cost_volume.size(1))`` whatever ``max_disp`` says
    (both branches at :24-28 reduce to that)."""

    def __init__(self, max_disp, match_similarity=True):
        super().__init__()
        self.max_disp = max_disp
        self.match_similarity = match_similarity

    def forward(self, cost_volume):
        return _soft_argmin(cost_volume, self.match_similarity, False, "DisparityEstimation")[0]


class DisparityEstimationWithProb(nn.Module):
    """``DisparityEstimationWithProb`` (estimation.py:37-62) -> (disp [B,H,W], prob [B,D,H,W])."""

    def __init__(self, max_disp, match_similarity=True):
        super().__init__()
        self.max_disp = max_disp
        self.match_similarity = match_similarity

    def forward(self, cost_volume):
        return _soft_argmin(cost_volume, self.match_similarity, True, "DisparityEstimationWithProb")


def disparity_regression(x, maxdisp):
    """``disparity_regression`` (submodule.py:23-27): sum_d x[:,d]*d for a probability volume x."""
    assert len(x.shape) == 4
    x = N.dev_f32(x, "disparity_regression: x")
    B, D, H, W = x.shape
    if D != maxdisp:
        raise RuntimeError(f"disparity_regression: x has {D} planes but maxdisp={maxdisp}")  # broadcast error in torch
    out = torch.empty((B, H, W), device=x.device, dtype=torch.float32)
    with torch.cuda.device(x.device):
        N.call("stereo_disparity_regression_f32", N.ptr(x), N.ptr(out), B, D, H, W, N.stream_ptr(x.device))
    return out


def group_mean_soft_argmin(volume_g, match_similarity=True, return_mean=False):
    """cfg-2 chain reader: ``DisparityEstimation(D)(volume_g.mean(1))`` for a group-wise
    volume [B,G,D,H,W] in ONE pass over it; optionally also returns the mean volume."""
    vg = N.dev_f32(volume_g, "group_mean_soft_argmin: volume_g")
    if vg.dim() != 5:
        raise ValueError("group_mean_soft_argmin: expected [B,G,D,H,W]")
    B, G, D, H, W = vg.shape
    disp = torch.empty((B, H, W), device=vg.device, dtype=torch.float32)
    mean = torch.empty((B, D, H, W), device=vg.device, dtype=torch.float32) if return_mean else None
    with torch.cuda.device(vg.device):
        N.call("stereo_group_mean_soft_argmin_f32", N.ptr(vg), N.ptr(mean), N.ptr(disp), B, G, D, H, W,
               1 if match_similarity else 0, N.stream_ptr(vg.device))
    return (disp, mean) if return_mean else disp


def corr_soft_argmin(left_feature, right_feature, max_disp, match_similarity=True):
    """Fused ``DisparityEstimation(D)(CostVolume(D,'correlation')(L,R))`` without
    materialising the volume (opt-in fast path; same numbers as the two-step chain)."""
    L = N.dev_f32(left_feature, "corr_soft_argmin: left_feature")
    R = N.dev_f32(right_feature, "corr_soft_argmin: right_feature")
    if L.dim() != 4 or L.shape != R.shape:
        raise ValueError("corr_soft_argmin: expected two [B,C,H,W] tensors of one shape")
    B, C, H, W = L.shape
    disp = torch.empty((B, H, W), device=L.device, dtype=torch.float32)
    with torch.cuda.device(L.device):
        N.call("stereo_corr_soft_argmin_f32", N.ptr(L), N.ptr(R), N.ptr(disp), B, C, H, W, int(max_disp),
               1 if match_similarity else 0, N.stream_ptr(L.device))
    return disp
