"""Soft-argmin disparity regression (A4) behind the reference's surfaces:
``models/utils/estimation.py:7-62`` and ``models/utils/submodule.py:23-27``."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _native as N
from . import ops


class _SoftArgminFn(torch.autograd.Function):
    """Differentiable soft-argmin (optionally with the probability volume as a second output)."""

    @staticmethod
    def forward(ctx, cost_volume, match_similarity, with_prob):
        disp, prob = _soft_argmin_fwd(cost_volume, match_similarity, with_prob, "DisparityEstimation")
        ctx.save_for_backward(cost_volume)
        ctx.cfg = (bool(match_similarity), bool(with_prob))
        if with_prob:
            return disp, prob
        return disp

    @staticmethod
    def backward(ctx, grad_disp, grad_prob=None):
        (vol,) = ctx.saved_tensors
        sim, with_prob = ctx.cfg
        B, D, H, W = vol.shape
        gd = N.dev_f32(grad_disp if grad_disp is not None else torch.zeros((B, H, W), device=vol.device),
                       "soft-argmin backward: grad_disp")
        gp = N.dev_f32(grad_prob, "soft-argmin backward: grad_prob") if (with_prob and grad_prob is not None) else None
        gvol = torch.empty_like(vol)
        with torch.cuda.device(vol.device):
            N.call("stereo_soft_argmin_bwd_f32", N.ptr(vol), N.ptr(gd), N.ptr(gp), N.ptr(gvol), B, D, H, W,
                   1 if sim else 0, N.stream_ptr(vol.device))
        return gvol, None, None


def _soft_argmin(cost_volume, match_similarity, with_prob, who):
    assert cost_volume.dim() == 4  # estimation.py:15
    if ops.tracing() and not with_prob:   # torch.compile / export: the registered op
        return torch.ops.stereo_hip.soft_argmin(cost_volume, bool(match_similarity)), None
    if torch.is_grad_enabled() and cost_volume.requires_grad:
        with torch.no_grad():
            N.dev_f32(cost_volume, f"{who}: cost_volume")  # validation only
        out = _SoftArgminFn.apply(cost_volume.contiguous(), bool(match_similarity), bool(with_prob))
        return out if with_prob else (out, None)
    return _soft_argmin_fwd(cost_volume, match_similarity, with_prob, who)


def _soft_argmin_fwd(cost_volume, match_similarity, with_prob, who):
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


class _GroupMeanSoftArgminFn(torch.autograd.Function):
    """Differentiable fused reader: d/d volume_g[:, g] = (soft-argmin backward on the mean volume + the mean
    volume's own cotangent) / G for every group."""

    @staticmethod
    def forward(ctx, volume_g, match_similarity):
        disp, mean = _group_mean_fwd(volume_g, match_similarity, True)
        ctx.save_for_backward(mean)
        ctx.cfg = (bool(match_similarity), volume_g.shape[1])
        return disp, mean

    @staticmethod
    def backward(ctx, grad_disp, grad_mean):
        (mean,) = ctx.saved_tensors
        sim, G = ctx.cfg
        B, D, H, W = mean.shape
        gd = N.dev_f32(grad_disp if grad_disp is not None else torch.zeros((B, H, W), device=mean.device),
                       "group_mean_soft_argmin backward: grad_disp")
        gmean = torch.empty_like(mean)
        with torch.cuda.device(mean.device):
            N.call("stereo_soft_argmin_bwd_f32", N.ptr(mean), N.ptr(gd), N.ptr(None), N.ptr(gmean), B, D, H, W,
                   1 if sim else 0, N.stream_ptr(mean.device))
        if grad_mean is not None:
            gmean = gmean + grad_mean
        return (gmean / G).unsqueeze(1).expand(B, G, D, H, W).contiguous(), None


def _group_mean_fwd(vg, match_similarity, return_mean):
    B, G, D, H, W = vg.shape
    disp = torch.empty((B, H, W), device=vg.device, dtype=torch.float32)
    mean = torch.empty((B, D, H, W), device=vg.device, dtype=torch.float32) if return_mean else None
    with torch.cuda.device(vg.device):
        N.call("stereo_group_mean_soft_argmin_f32", N.ptr(vg), N.ptr(mean), N.ptr(disp), B, G, D, H, W,
               1 if match_similarity else 0, N.stream_ptr(vg.device))
    return disp, mean


def group_mean_soft_argmin(volume_g, match_similarity=True, return_mean=False):
    """cfg-2 chain reader: ``DisparityEstimation(D)(volume_g.mean(1))`` for a group-wise
    volume [B,G,D,H,W] in ONE pass over it; optionally also returns the mean volume."""
    if volume_g.dim() != 5:
        raise ValueError("group_mean_soft_argmin: expected [B,G,D,H,W]")
    if torch.is_grad_enabled() and volume_g.requires_grad:
        with torch.no_grad():
            N.dev_f32(volume_g, "group_mean_soft_argmin: volume_g")  # validation only
        disp, mean = _GroupMeanSoftArgminFn.apply(volume_g.contiguous(), bool(match_similarity))
        return (disp, mean) if return_mean else disp
    vg = N.dev_f32(volume_g, "group_mean_soft_argmin: volume_g")
    disp, mean = _group_mean_fwd(vg, match_similarity, return_mean)
    return (disp, mean) if return_mean else disp


def corr_soft_argmin(left_feature, right_feature, max_disp, match_similarity=True):
    """Fused ``DisparityEstimation(D)(CostVolume(D,'correlation')(L,R))`` without
    materialising the volume (opt-in fast path; same numbers as the two-step chain)."""
    L = N.dev_feat(left_feature, "corr_soft_argmin: left_feature")   # fp32 or fp16 features
    R = N.dev_feat(right_feature, "corr_soft_argmin: right_feature")
    if L.dtype != R.dtype:
        L, R = L.float(), R.float()
    if L.dim() != 4 or L.shape != R.shape:
        raise ValueError("corr_soft_argmin: expected two [B,C,H,W] tensors of one shape")
    B, C, H, W = L.shape
    disp = torch.empty((B, H, W), device=L.device, dtype=torch.float32)
    with torch.cuda.device(L.device):
        N.feat_call("stereo_corr_soft_argmin", [L, R],
                    lambda f: (N.ptr(f[0]), N.ptr(f[1]), N.ptr(disp), B, C, H, W, int(max_disp),
                               1 if match_similarity else 0, N.stream_ptr(L.device)))
    return disp
