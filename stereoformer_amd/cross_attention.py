"""Epipolar cross-attention cost volume (A7).

The reference file ``models/CrossAttentionCostVolume/cross_attention_cost_volume.py`` is
EMPTY (0 bytes): this module is build-defined and parity-unpinned by the reference; it is
pinned by identities to ``CostVolume('correlation')`` + ``DisparityEstimation`` (mask='zero',
scale='mean') and to ``CorrBlock1D.corr`` (scale='sqrt_c') -- SURVEY.md 8(c) I2/I3."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _native as N
from . import ops

class _BandApplyFn(torch.autograd.Function):
    """``out[b,c,h,w] = sum_{d <= w} P[b,d,h,w] * V[b,c,h,w-d]`` -- the P V half of the attention as the transpose of the
    correlation volume: it IS A1's backward w.r.t. the left features (times Cv), and its own backward is A1's forward
    (d/dP) and A1's backward w.r.t. the right features (d/dV).  All three run on the HIP volume kernels."""

    @staticmethod
    def forward(ctx, P, V):
        B, D, H, W = P.shape
        Cv = V.shape[1]
        out = torch.empty((B, Cv, H, W), device=P.device, dtype=torch.float32)
        with torch.cuda.device(P.device):
            N.call("stereo_gwc_volume_bwd_f32", N.ptr(V), N.ptr(V), N.ptr(P), N.ptr(out), N.ptr(None), B, Cv, H, W, D, 1,
                   N.stream_ptr(P.device))
        ctx.save_for_backward(P, V)
        return out.mul_(float(Cv))

    @staticmethod
    def backward(ctx, grad_out):
        P, V = ctx.saved_tensors
        B, D, H, W = P.shape
        Cv = V.shape[1]
        go = N.dev_f32(grad_out, "cross-attention backward: grad_out")
        gP = gV = None
        with torch.cuda.device(P.device):
            if ctx.needs_input_grad[0]:
                gP = torch.empty_like(P)
                N.call("stereo_corr_volume_f32", N.ptr(go), N.ptr(V), N.ptr(gP), B, Cv, H, W, D, N.stream_ptr(P.device))
                gP.mul_(float(Cv))
            if ctx.needs_input_grad[1]:
                gV = torch.empty_like(V)
                N.call("stereo_gwc_volume_bwd_f32", N.ptr(go), N.ptr(V), N.ptr(P), N.ptr(None), N.ptr(gV), B, Cv, H, W, D, 1,
                       N.stream_ptr(P.device))
                gV.mul_(float(Cv))
        return gP, gV


_MASK = {"zero": 0, "neginf": 1}
_SCALE = {"sqrt_c": 0, "mean": 1}


class CrossAttentionCostVolume(nn.Module):
    """``CrossAttentionCostVolume(max_disp, mask='zero'|'neginf', scale='sqrt_c'|'mean')``.

    forward(left_feature, right_feature[, value]) -> (out [B,Cv,H,W], disp [B,H,W]):
    per row, logits ``<L[:,w], R[:,w-d]> * scale`` for d in [0, max_disp), softmax over d,
    ``disp = sum_d p_d d`` and ``out[:,w] = sum_d p_d V[:,w-d]`` (V defaults to the right features).

    Inference kernels are fp32-faithful on split fp16 / bf16 matrix-core operands; the default ones for C <= 32 with V = right
    features and for C > 32 with ``need_out=False`` (max_disp <= 128) assume ``|feature| < 4094`` (fp16 features: 65504) and
    return inf / NaN beyond that, never a silently wrong number (include/stereo_hip.h, A7)."""

    def __init__(self, max_disp, mask="zero", scale="sqrt_c"):
        super().__init__()
        if mask not in _MASK or scale not in _SCALE:
            raise ValueError("CrossAttentionCostVolume: mask in {'zero','neginf'}, scale in {'sqrt_c','mean'}")
        self.max_disp = max_disp
        self.mask = mask
        self.scale = scale

    def _forward_differentiable(self, L, R, V, need_out):
        """Training path (grad mode): the same function as the fused inference kernel, composed of differentiable HIP ops --
        A1 forward / backward for Q K^T (``CostVolume``) and for P V (``_BandApplyFn``), softmax and the disparity expectation
        as elementwise torch ops on the [B,D,H,W] score tensor (materialised here, unlike at inference)."""
        from .cost_volume import CostVolume
        L, R, V = L.float().contiguous(), R.float().contiguous(), V.float().contiguous()
        B, C, H, W = L.shape
        D = int(self.max_disp)
        vol = CostVolume(D, "correlation")(L, R)                        # mean over channels, 0 where w < d
        logits = vol * (1.0 if self.scale == "mean" else float(C) ** 0.5)  # x C x scale
        if self.mask == "neginf":
            d = torch.arange(D, device=L.device).view(1, D, 1, 1)
            w = torch.arange(W, device=L.device).view(1, 1, 1, W)
            logits = logits.masked_fill(w < d, float("-inf"))
        P = torch.softmax(logits, dim=1)
        disp = (P * torch.arange(D, device=L.device, dtype=torch.float32).view(1, D, 1, 1)).sum(1)
        out = _BandApplyFn.apply(P.contiguous(), V) if need_out else None
        return out, disp

    def forward(self, left_feature, right_feature, value=None, need_out=True):
        if ops.tracing():   # torch.compile / export: the registered (inference) op
            out, disp = torch.ops.stereo_hip.cross_attention(left_feature, right_feature,
                                                             right_feature if value is None else value, int(self.max_disp),
                                                             _MASK[self.mask], _SCALE[self.scale])
            return (out if need_out else None), disp
        if torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad
                                           for t in (left_feature, right_feature, value)):
            if left_feature.dim() != 4 or left_feature.shape != right_feature.shape:
                raise ValueError("CrossAttentionCostVolume: expected two [B,C,H,W] tensors of one shape")
            return self._forward_differentiable(left_feature, right_feature,
                                                right_feature if value is None else value, need_out)
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
