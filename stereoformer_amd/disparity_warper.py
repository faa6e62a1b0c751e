"""``disp_warp`` (A12) behind ``utils/disparity_warper.py:83-106``."""
from __future__ import annotations

import torch

from . import _native as N


class _DispWarpFn(torch.autograd.Function):
    """Differentiable w.r.t. the disparity (the image is a model input); the validity mask carries no gradient."""

    @staticmethod
    def forward(ctx, img, disp, padding_mode):
        warped, mask = disp_warp(img, disp, padding_mode)
        ctx.save_for_backward(img, disp)
        ctx.border = padding_mode == "border"
        ctx.mark_non_differentiable(mask)
        return warped, mask

    @staticmethod
    def backward(ctx, grad_warped, _grad_mask):
        img, disp = ctx.saved_tensors
        b, c, h, w = img.shape
        gw = N.dev_f32(grad_warped, "disp_warp backward: grad")
        gdisp = torch.empty_like(disp)
        with torch.cuda.device(img.device):
            N.call("stereo_disp_warp_bwd_disp_f32", N.ptr(img), N.ptr(disp), N.ptr(gw), N.ptr(gdisp), b, c, h, w,
                   1 if ctx.border else 0, N.stream_ptr(img.device))
        return None, gdisp, None


def disp_warp(img, disp, padding_mode="border", check_positive=False):
    """Warp ``img`` [B,C,H,W] horizontally by ``-disp`` [B,1,H,W]; returns
    (warped, valid_mask), both [B,C,H,W].

    The reference asserts ``disp.min() >= 0`` (:93), a host sync inside the GRU
    loop; it is opt-in here (``check_positive=True``) so the loop stays
    capturable.  The kernel itself is defined for any disp."""
    if torch.is_grad_enabled() and disp.requires_grad and not img.requires_grad:
        if check_positive:
            assert disp.min() >= 0
        return _DispWarpFn.apply(img.contiguous(), disp.contiguous(), padding_mode)
    img = N.dev_f32(img, "disp_warp: img")
    disp = N.dev_f32(disp, "disp_warp: disp")
    b, c, h, w = img.shape
    if tuple(disp.shape) != (b, 1, h, w):
        raise ValueError("disp_warp: disp must be [B,1,H,W]")
    if padding_mode not in ("border", "zeros"):
        raise ValueError("disp_warp: padding_mode must be 'border' or 'zeros'")
    if check_positive:
        assert disp.min() >= 0
    warped = torch.empty_like(img)
    mask = torch.empty_like(img)
    with torch.cuda.device(img.device):
        N.call("stereo_disp_warp_f32", N.ptr(img), N.ptr(disp), N.ptr(warped), N.ptr(mask), b, c, h, w,
               1 if padding_mode == "border" else 0, N.stream_ptr(img.device))
    return warped, mask
