"""Convex / bilinear upsampling (A13) behind ``models/utils/upsample.py``,
``RAFT_Stereo.upsample_flow`` (raft_stereo.py:110-122) and ``upflow8`` (utils/utils.py:85-87)."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _native as N


class ConvAffinityUpsample(nn.Module):
    """``ConvAffinityUpsample`` (upsample.py:6-19): the 0.25-scaled 576-channel mask head.
    The two convolutions stay with the host framework (MIOpen)."""

    def __init__(self, input_channels, hidden_channels=128):
        super().__init__()
        self.upsample_mask = nn.Sequential(
            nn.Conv2d(in_channels=input_channels, out_channels=hidden_channels, kernel_size=3, padding=1),
            nn.ReLU(inplace=True),
            nn.Conv2d(hidden_channels, 8 * 8 * 9, 1, padding=0))

    def forward(self, feature):
        return .25 * self.upsample_mask(feature)


class _UpsampleConvexFn(torch.autograd.Function):
    """Differentiable convex upsampling (factors 4 and 8): HIP forward + HIP backward."""

    @staticmethod
    def forward(ctx, disp, mask, factor):
        out = upsample_convex(disp, mask, factor)
        ctx.save_for_backward(disp, mask)
        ctx.factor = int(factor)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        disp, mask = ctx.saved_tensors
        n, cd, h, w = disp.shape
        go = N.dev_f32(grad_out, "upsample_convex backward: grad")
        gdisp = torch.zeros_like(disp)       # neighbour contributions are accumulated into it
        gmask = torch.empty_like(mask)
        with torch.cuda.device(disp.device):
            N.call("stereo_upsample_convex_bwd_f32", N.ptr(disp), N.ptr(mask), N.ptr(go), N.ptr(gdisp), N.ptr(gmask), n,
                   cd, h, w, ctx.factor, N.stream_ptr(disp.device))
        return gdisp, gmask, None


def upsample_convex(disp, mask, factor):
    """Convex-combination upsampling by ``factor``: disp [N,Cd,H,W], mask [N,9*f*f,H,W]."""
    if torch.is_grad_enabled() and (disp.requires_grad or mask.requires_grad):
        if factor not in (4, 8):
            raise RuntimeError("upsample_convex: backward is built for factors 4 and 8 only")
        return _UpsampleConvexFn.apply(disp.contiguous(), mask.contiguous(), int(factor))
    disp = N.dev_f32(disp, "upsample_convex: disp")
    n, cd, h, w = disp.shape
    if tuple(mask.shape) != (n, 9 * factor * factor, h, w):
        raise RuntimeError(f"upsample_convex: mask shape {tuple(mask.shape)} is not "
                           f"[{n},{9 * factor * factor},{h},{w}]")  # the reference's .view would fail
    out = torch.empty((n, cd, factor * h, factor * w), device=disp.device, dtype=torch.float32)
    # a channels-last mask (the update blocks' mask heads on the channels-last convolution kernels) is read as it is
    if (factor in (4, 8) and mask.is_cuda and mask.dtype == torch.float32 and not mask.is_contiguous()
            and mask.is_contiguous(memory_format=torch.channels_last)):
        with torch.cuda.device(disp.device):
            try:
                N.call("stereo_upsample_convex_nhwc_mask_f32", N.ptr(disp), N.ptr(mask), N.ptr(out), n, cd, h, w, int(factor),
                       N.stream_ptr(disp.device))
                return out
            except N.StereoHipError as e:
                if e.code != N.ERR_UNSUPPORTED:
                    raise
    mask = N.dev_f32(mask, "upsample_convex: mask")
    with torch.cuda.device(disp.device):
        N.call("stereo_upsample_convex_f32", N.ptr(disp), N.ptr(mask), N.ptr(out), n, cd, h, w, int(factor),
               N.stream_ptr(disp.device))
    return out


def upsample_convex8(disp, mask):
    """``upsample_convex8`` (upsample.py:22-38): [N,1,H,W] x [N,576,H,W] -> [N,1,8H,8W]."""
    if disp.shape[1] != 1:
        raise RuntimeError("upsample_convex8: disp must have one channel")
    return upsample_convex(disp, mask, 8)


def upsample_flow(flow, mask, downsample=2):
    """``RAFT_Stereo.upsample_flow`` (raft_stereo.py:110-122) with factor 2**downsample."""
    return upsample_convex(flow, mask, 2 ** downsample)


class _ResizeBilinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, size, align_corners, scale):
        ctx.cfg = (tuple(x.shape), (int(size[0]), int(size[1])), bool(align_corners), float(scale))
        return resize_bilinear(x, size, align_corners, scale)

    @staticmethod
    def backward(ctx, grad_out):
        (n, c, hi, wi), (ho, wo), ac, scale = ctx.cfg
        go = N.dev_f32(grad_out, "resize_bilinear backward: grad")
        gin = torch.zeros((n, c, hi, wi), device=go.device, dtype=torch.float32)
        with torch.cuda.device(go.device):
            N.call("stereo_resize_bilinear_bwd_f32", N.ptr(go), N.ptr(gin), n * c, hi, wi, ho, wo, 1 if ac else 0, scale,
                   N.stream_ptr(go.device))
        return gin, None, None, None


def resize_bilinear(x, size, align_corners, scale=1.0):
    """``scale * F.interpolate(x, size, mode='bilinear', align_corners=...)`` in one kernel (HIP backward)."""
    if torch.is_grad_enabled() and x.requires_grad:
        return _ResizeBilinearFn.apply(x.contiguous(), (int(size[0]), int(size[1])), bool(align_corners), float(scale))
    ho, wo = int(size[0]), int(size[1])
    if (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] % 4 == 0
            and not x.is_contiguous() and x.is_contiguous(memory_format=torch.channels_last)):
        # channels-last in, channels-last out (the update block in torch.channels_last): no layout round trip
        n, c, hi, wi = x.shape
        out = torch.empty((n, c, ho, wo), device=x.device, dtype=torch.float32, memory_format=torch.channels_last)
        with torch.cuda.device(x.device):
            N.call("stereo_resize_bilinear_nhwc_f32", N.ptr(x), N.ptr(out), n, c, hi, wi, ho, wo,
                   1 if align_corners else 0, float(scale), N.stream_ptr(x.device))
        return out
    x = N.dev_f32(x, "resize_bilinear: x")
    n, c, hi, wi = x.shape
    out = torch.empty((n, c, ho, wo), device=x.device, dtype=torch.float32)
    with torch.cuda.device(x.device):
        N.call("stereo_resize_bilinear_f32", N.ptr(x), N.ptr(out), n * c, hi, wi, ho, wo,
               1 if align_corners else 0, float(scale), N.stream_ptr(x.device))
    return out


def upsample_simple8(disp, mode="bilinear"):
    """``upsample_simple8`` (upsample.py:41-43): 8 * bilinear x8, align_corners=True."""
    if mode != "bilinear":
        raise NotImplementedError
    return resize_bilinear(disp, (8 * disp.shape[2], 8 * disp.shape[3]), True, 8.0)


def upflow8(flow, mode="bilinear"):
    """``upflow8`` (utils/utils.py:85-87)."""
    return upsample_simple8(flow, mode)


def interp(x, dest):
    """``interp`` (update/update.py:125-127): bilinear, align_corners=True, to dest's size."""
    return resize_bilinear(x, dest.shape[2:], True)
