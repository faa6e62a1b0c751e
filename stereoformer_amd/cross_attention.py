"""Epipolar cross-attention cost volume (A7).

The reference file ``models/CrossAttentionCostVolume/cross_attention_cost_volume.py`` is
EMPTY (0 bytes): this module is build-defined and parity-unpinned by the reference; it is
pinned by identities to ``CostVolume('correlation')`` + ``DisparityEstimation`` (mask='zero',
scale='mean') and to ``CorrBlock1D.corr`` (scale='sqrt_c') -- SURVEY.md 8(c) I2/I3."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _native as N

_MASK = {"zero": 0, "neginf": 1}
_SCALE = {"sqrt_c": 0, "mean": 1}


class CrossAttentionCostVolume(nn.Module):
    """``CrossAttentionCostVolume(max_disp, mask='zero'|'neginf', scale='sqrt_c'|'mean')``.

    forward(left_feature, right_feature[, value]) -> (out [B,Cv,H,W], disp [B,H,W]):
    per row, logits ``<L[:,w], R[:,w-d]> * scale`` for d in [0, max_disp), softmax over d,
    ``disp = sum_d p_d d`` and ``out[:,w] = sum_d p_d V[:,w-d]`` (V defaults to the right features)."""

    def __init__(self, max_disp, mask="zero", scale="sqrt_c"):
        super().__init__()
        if mask not in _MASK or scale not in _SCALE:
            raise ValueError("CrossAttentionCostVolume: mask in {'zero','neginf'}, scale in {'sqrt_c','mean'}")
        self.max_disp = max_disp
        self.mask = mask
        self.scale = scale

    def forward(self, left_feature, right_feature, value=None, need_out=True):
        # features fp32 or fp16 (widened on the way into LDS, f32 accumulate, fp32 outputs)
        L = N.dev_feat(left_feature, "CrossAttentionCostVolume: left_feature")
        R = N.dev_feat(right_feature, "CrossAttentionCostVolume: right_feature")
        if L.dim() != 4 or L.shape != R.shape:
            raise ValueError("CrossAttentionCostVolume: expected two [B,C,H,W] tensors of one shape")
        V = R if value is None else N.dev_feat(value, "CrossAttentionCostVolume: value")
        if not (L.dtype == R.dtype == V.dtype):     # mixed precisions: the fp32 path
            same = V is R
            L, R = L.float(), R.float()
            V = R if same else V.float()
        B, C, H, W = L.shape
        if V.dim() != 4 or V.shape[0] != B or tuple(V.shape[2:]) != (H, W):
            raise ValueError("CrossAttentionCostVolume: value must be [B,Cv,H,W]")
        Cv = V.shape[1]
        out = torch.empty((B, Cv, H, W), device=L.device, dtype=torch.float32) if need_out else None
        disp = torch.empty((B, H, W), device=L.device, dtype=torch.float32)
        same = V is R
        with torch.cuda.device(L.device):
            N.feat_call("stereo_cross_attention", [L, R, None if same else V],
                        lambda f: (N.ptr(f[0]), N.ptr(f[1]), N.ptr(f[1] if same else f[2]), N.ptr(out), N.ptr(disp), B, C, Cv,
                                   H, W, int(self.max_disp), _MASK[self.mask], _SCALE[self.scale], N.stream_ptr(L.device)))
        return out, disp
