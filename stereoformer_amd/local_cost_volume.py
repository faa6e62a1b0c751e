"""Local cost-volume re-sampling (A5/A6) behind the reference's surfaces:
``models/LocalCostVolume/Attempts/fixed_local_cost_volume.py:13-80`` and
``models/LocalCostVolume/Attempts/variance_local_cost_volume.py:12-99``."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _native as N
from . import ops


def _vol_disp(old_cost_volume, cur_disparity, who):
    vol = N.dev_f32(old_cost_volume, f"{who}: old_cost_volume")
    disp = N.dev_f32(cur_disparity, f"{who}: cur_disparity")
    if vol.dim() != 4:
        raise ValueError(f"{who}: old_cost_volume must be [B,D,H,W]")
    B, D, H, W = vol.shape
    if tuple(disp.shape) != (B, 1, H, W):
        raise ValueError(f"{who}: cur_disparity must be [B,1,H,W] = {(B, 1, H, W)}, got {tuple(disp.shape)}")
    N.same_device(vol, disp)
    return vol, disp


class _LocalLookupFn(torch.autograd.Function):
    """Differentiable fixed-radius local lookup (A5) with the reference's STE semantics."""

    @staticmethod
    def forward(ctx, vol, disp, radius, sample_points, consider_valid):
        out = _local_lookup_fwd(vol, disp, radius, sample_points, consider_valid)
        ctx.save_for_backward(vol, disp)
        ctx.cfg = (float(radius), int(sample_points), bool(consider_valid))
        return out

    @staticmethod
    def backward(ctx, grad_out):
        vol, disp = ctx.saved_tensors
        radius, S, cv = ctx.cfg
        B, D, H, W = vol.shape
        go = N.dev_f32(grad_out, "LocalCostVolume backward: grad")
        gdisp = torch.empty_like(disp)
        with torch.cuda.device(vol.device):
            # the dense form writes every element of the gradient volume itself (one pass at the store rate); wide radii /
            # ragged H*W take the zero-fill + accumulate form
            gvol = torch.empty_like(vol)
            try:
                N.call("stereo_local_lookup_bwd_dense_f32", N.ptr(vol), N.ptr(disp), N.ptr(go), N.ptr(gvol), N.ptr(gdisp), B, D,
                       H, W, radius, S, 1 if cv else 0, N.stream_ptr(vol.device))
            except N.StereoHipError as e:
                if e.code != N.ERR_UNSUPPORTED:
                    raise
                gvol.zero_()
                N.call("stereo_local_lookup_bwd_f32", N.ptr(vol), N.ptr(disp), N.ptr(go), N.ptr(gvol), N.ptr(gdisp), B, D, H,
                       W, radius, S, 1 if cv else 0, N.stream_ptr(vol.device))
        return gvol, gdisp, None, None, None


def _local_lookup_fwd(vol, disp, radius, sample_points, consider_valid):
    B, D, H, W = vol.shape
    out = torch.empty((B, 1, H, W), device=vol.device, dtype=torch.float32)
    with torch.cuda.device(vol.device):
        N.call("stereo_local_lookup_f32", N.ptr(vol), N.ptr(disp), N.ptr(out), B, D, H, W, float(radius),
               int(sample_points), 1 if consider_valid else 0, N.stream_ptr(vol.device))
    return out


class LocalCostVolume(nn.Module):
    """``LocalCostVolume(radius=4, sample_points=10)`` (fixed_local_cost_volume.py:13-58)."""

    def __init__(self, radius=4, sample_points=10) -> None:
        super().__init__()
        self.radius = radius
        self.sample_points = sample_points

    def forward(self, old_cost_volume, cur_disparity, consider_valid=False):
        if ops.tracing():   # torch.compile / export: the registered op
            return torch.ops.stereo_hip.local_lookup(old_cost_volume, cur_disparity, float(self.radius),
                                                     int(self.sample_points), bool(consider_valid))
        if torch.is_grad_enabled() and (old_cost_volume.requires_grad or cur_disparity.requires_grad):
            with torch.no_grad():
                _vol_disp(old_cost_volume, cur_disparity, "LocalCostVolume")  # validation only
            return _LocalLookupFn.apply(old_cost_volume.contiguous(), cur_disparity.contiguous(), float(self.radius),
                                        int(self.sample_points), bool(consider_valid))
        vol, disp = _vol_disp(old_cost_volume, cur_disparity, "LocalCostVolume")
        return _local_lookup_fwd(vol, disp, self.radius, self.sample_points, consider_valid)


def _variance_lookup_fwd(vol, disp, gamma, sample_points, consider_valid):
    B, D, H, W = vol.shape
    out = torch.empty((B, 1, H, W), device=vol.device, dtype=torch.float32)
    with torch.cuda.device(vol.device):
        N.call("stereo_variance_lookup_f32", N.ptr(vol), N.ptr(disp), N.ptr(out), B, D, H, W, float(gamma),
               int(sample_points), 1 if consider_valid else 0, N.stream_ptr(vol.device))
    return out


class _VarianceLookupFn(torch.autograd.Function):
    """Differentiable variance-based lookup (A6): HIP forward + HIP backward that follows the reference's autograd graph
    (softmax variance -> bounds -> candidates -> gathered taps -> softmax expectation; ``.long()`` indices carry no
    gradient, the interpolation rates do)."""

    @staticmethod
    def forward(ctx, vol, disp, gamma, sample_points, consider_valid):
        out = _variance_lookup_fwd(vol, disp, gamma, sample_points, consider_valid)
        ctx.save_for_backward(vol, disp)
        ctx.cfg = (float(gamma), int(sample_points), bool(consider_valid))
        return out

    @staticmethod
    def backward(ctx, grad_out):
        vol, disp = ctx.saved_tensors
        gamma, S, cv = ctx.cfg
        B, D, H, W = vol.shape
        go = N.dev_f32(grad_out, "Variance_Based_Local_Cost_Volume backward: grad")
        gvol = torch.empty_like(vol)
        gdisp = torch.empty_like(disp)
        with torch.cuda.device(vol.device):
            try:   # one dense launch: every element of grad_vol written once (D <= 192, H*W % 4 == 0, sample_points <= 20)
                N.call("stereo_variance_lookup_bwd_dense_f32", N.ptr(vol), N.ptr(disp), N.ptr(go), N.ptr(gvol), N.ptr(gdisp), B,
                       D, H, W, gamma, S, 1 if cv else 0, N.stream_ptr(vol.device))
            except N.StereoHipError as e:
                if e.code != N.ERR_UNSUPPORTED:
                    raise
                gvol.zero_()           # taps and the variance path are accumulated into it
                N.call("stereo_variance_lookup_bwd_f32", N.ptr(vol), N.ptr(disp), N.ptr(go), N.ptr(gvol), N.ptr(gdisp), B, D,
                       H, W, gamma, S, 1 if cv else 0, N.stream_ptr(vol.device))
        return gvol, gdisp, None, None, None


class Variance_Based_Local_Cost_Volume(nn.Module):
    """``Variance_Based_Local_Cost_Volume(garma=1, sample_points=10)``
    (variance_local_cost_volume.py:28-78)."""

    def __init__(self, garma=1, sample_points=10):
        super().__init__()
        self.sample_points = sample_points
        self.garma = garma

    def forward(self, old_cost_volume, cur_disparity, consider_valid=False):
        if torch.is_grad_enabled() and (old_cost_volume.requires_grad or cur_disparity.requires_grad):
            with torch.no_grad():
                _vol_disp(old_cost_volume, cur_disparity, "Variance_Based_Local_Cost_Volume")  # validation only
            return _VarianceLookupFn.apply(old_cost_volume.contiguous(), cur_disparity.contiguous(), float(self.garma),
                                           int(self.sample_points), bool(consider_valid))
        vol, disp = _vol_disp(old_cost_volume, cur_disparity, "Variance_Based_Local_Cost_Volume")
        return _variance_lookup_fwd(vol, disp, self.garma, self.sample_points, consider_valid)


def build_cost_volume_from_volume(old_volume, sampling_candidates):
    """``build_cost_volume_from_volume`` (fixed_local_cost_volume.py:61-80):
    [B,D,H,W] x candidates [B,K,H,W] -> [B,K,H,W]."""
    vol = N.dev_f32(old_volume, "build_cost_volume_from_volume: old_volume")
    cand = N.dev_f32(sampling_candidates, "build_cost_volume_from_volume: sampling_candidates")
    B, D, H, W = vol.shape
    if cand.dim() != 4 or cand.shape[0] != B or tuple(cand.shape[2:]) != (H, W):
        raise ValueError("build_cost_volume_from_volume: candidates must be [B,K,H,W]")
    out = torch.empty_like(cand)
    with torch.cuda.device(vol.device):
        N.call("stereo_sample_volume_f32", N.ptr(vol), N.ptr(cand), N.ptr(out), B, D, H, W, int(cand.shape[1]),
               N.stream_ptr(vol.device))
    return out


class _BoundsLookupFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, vol, disp, bounds, sample_points, consider_valid):
        ctx.save_for_backward(vol, disp, bounds)
        ctx.cfg = (int(sample_points), bool(consider_valid))
        return bounds_lookup(vol, disp, bounds, sample_points, consider_valid)

    @staticmethod
    def backward(ctx, grad_out):
        vol, disp, bounds = ctx.saved_tensors
        S, cv = ctx.cfg
        B, D, H, W = vol.shape
        go = N.dev_f32(grad_out, "bounds_lookup backward: grad")
        gvol, gdisp, gb = torch.zeros_like(vol), torch.empty_like(disp), torch.empty_like(bounds)
        with torch.cuda.device(vol.device):
            N.call("stereo_bounds_lookup_bwd_f32", N.ptr(vol), N.ptr(disp), N.ptr(bounds), N.ptr(go), N.ptr(gvol),
                   N.ptr(gdisp), N.ptr(gb), B, D, H, W, S, 1 if cv else 0, N.stream_ptr(vol.device))
        return gvol, gdisp, gb, None, None


class _UncertaintyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, vol, disp):
        ctx.save_for_backward(vol, disp)
        return uncertainty_volume(vol, disp)

    @staticmethod
    def backward(ctx, grad_u):
        vol, disp = ctx.saved_tensors
        B, D, H, W = vol.shape
        gu = N.dev_f32(grad_u, "uncertainty_volume backward: grad")
        gvol, gdisp = torch.empty_like(vol), torch.empty_like(disp)
        with torch.cuda.device(vol.device):
            N.call("stereo_uncertainty_volume_bwd_f32", N.ptr(vol), N.ptr(disp), N.ptr(gu), N.ptr(gvol), N.ptr(gdisp), B,
                   D, H, W, N.stream_ptr(vol.device))
        return gvol, gdisp


def bounds_lookup(old_cost_volume, cur_disp, bounds, sample_points, consider_valid=True):
    """The A11 sampler (``Attempts/update.py:279-313``): ``bounds`` [B,2,H,W] are the
    non-negative (lower, upper) offsets predicted by OffsetPredictionHead."""
    if torch.is_grad_enabled() and (old_cost_volume.requires_grad or cur_disp.requires_grad or bounds.requires_grad):
        return _BoundsLookupFn.apply(old_cost_volume.contiguous(), cur_disp.contiguous(), bounds.contiguous(),
                                     int(sample_points), bool(consider_valid))
    vol, disp = _vol_disp(old_cost_volume, cur_disp, "bounds_lookup")
    bounds = N.dev_f32(bounds, "bounds_lookup: bounds")
    B, D, H, W = vol.shape
    if tuple(bounds.shape) != (B, 2, H, W):
        raise ValueError("bounds_lookup: bounds must be [B,2,H,W]")
    out = torch.empty((B, 1, H, W), device=vol.device, dtype=torch.float32)
    with torch.cuda.device(vol.device):
        N.call("stereo_bounds_lookup_f32", N.ptr(vol), N.ptr(disp), N.ptr(bounds), N.ptr(out), B, D, H, W,
               int(sample_points), 1 if consider_valid else 0, N.stream_ptr(vol.device))
    return out


def uncertainty_volume(old_cost_volume, cur_disp):
    """``softmax(vol,1) * d * (d-disp)^2`` (``Attempts/update.py:71-72,86``) in one kernel."""
    if torch.is_grad_enabled() and (old_cost_volume.requires_grad or cur_disp.requires_grad):
        return _UncertaintyFn.apply(old_cost_volume.contiguous(), cur_disp.contiguous())
    vol, disp = _vol_disp(old_cost_volume, cur_disp, "uncertainty_volume")
    B, D, H, W = vol.shape
    out = torch.empty_like(vol)
    with torch.cuda.device(vol.device):
        N.call("stereo_uncertainty_volume_f32", N.ptr(vol), N.ptr(disp), N.ptr(out), B, D, H, W,
               N.stream_ptr(vol.device))
    return out
