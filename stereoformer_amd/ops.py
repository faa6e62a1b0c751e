"""Registered operators: the hot-path kernels as ``torch.ops.stereo_hip.*``.

The reference's style precedent is a pybind ``TORCH_LIBRARY``-like module for DCN (``deform/src/vision.cpp:5-12``).  Here every
kernel already sits behind the C ABI (``include/stereo_hip.h``); this module registers the main forward ops with
``torch.library`` -- schema, device (HIP) implementation = the C-ABI call, FakeTensor (meta) shape function, and the autograd
formula wired to the HIP backward kernels -- so that models built from the module surfaces are traceable by
``torch.compile`` / ``torch.export`` (``fullgraph=True``: no graph break at the ctypes boundary).  The module classes
(``CostVolume`` ...) route through these ops while a compiler is tracing and call the C ABI directly otherwise.

There is still no CPU implementation: the ops are registered for the HIP device only ("cuda" dispatch key on ROCm builds).
"""
from __future__ import annotations

import torch

from . import _native as N

def _st(t):
    return N.stream_ptr(t.device)


# ---------------------------------------------------------------------------------------------- A1 / A3 volumes
@torch.library.custom_op("stereo_hip::gwc_volume", mutates_args=(), device_types="cuda")
def gwc_volume(left: torch.Tensor, right: torch.Tensor, max_disp: int, num_groups: int) -> torch.Tensor:
    """A3 (A1 for num_groups = 1): [B,C,H,W] x 2 -> [B,G,D,H,W]."""
    L, R = left.contiguous(), right.contiguous()
    B, C, H, W = L.shape
    vol = torch.empty((B, num_groups, max_disp, H, W), device=L.device, dtype=torch.float32)
    with torch.cuda.device(L.device):
        N.feat_call("stereo_gwc_volume", [L, R],
                    lambda f: (N.ptr(f[0]), N.ptr(f[1]), N.ptr(vol), B, C, H, W, int(max_disp), int(num_groups), _st(L)))
    return vol


@gwc_volume.register_fake
def _(left, right, max_disp, num_groups):
    B, C, H, W = left.shape
    torch._check(C % num_groups == 0, lambda: "C % num_groups != 0")
    return left.new_empty((B, num_groups, max_disp, H, W), dtype=torch.float32)


@torch.library.custom_op("stereo_hip::gwc_volume_bwd", mutates_args=(), device_types="cuda")
def gwc_volume_bwd(left: torch.Tensor, right: torch.Tensor, grad_vol: torch.Tensor, max_disp: int,
                   num_groups: int) -> tuple[torch.Tensor, torch.Tensor]:
    L, R, g = left.float().contiguous(), right.float().contiguous(), grad_vol.contiguous()
    B, C, H, W = L.shape
    gl, gr = torch.empty_like(L), torch.empty_like(R)
    with torch.cuda.device(L.device):
        N.call("stereo_gwc_volume_bwd_f32", N.ptr(L), N.ptr(R), N.ptr(g), N.ptr(gl), N.ptr(gr), B, C, H, W, int(max_disp),
               int(num_groups), _st(L))
    return gl, gr


@gwc_volume_bwd.register_fake
def _(left, right, grad_vol, max_disp, num_groups):
    return left.new_empty(left.shape, dtype=torch.float32), right.new_empty(right.shape, dtype=torch.float32)


def _gwc_setup(ctx, inputs, output):
    left, right, ctx.max_disp, ctx.num_groups = inputs
    ctx.save_for_backward(left, right)


def _gwc_backward(ctx, grad):
    left, right = ctx.saved_tensors
    gl, gr = gwc_volume_bwd(left, right, grad, ctx.max_disp, ctx.num_groups)
    return gl.to(left.dtype), gr.to(right.dtype), None, None


gwc_volume.register_autograd(_gwc_backward, setup_context=_gwc_setup)


# ---------------------------------------------------------------------------------------------- A4 soft-argmin
@torch.library.custom_op("stereo_hip::soft_argmin", mutates_args=(), device_types="cuda")
def soft_argmin(cost_volume: torch.Tensor, match_similarity: bool) -> torch.Tensor:
    """A4: [B,D,H,W] -> [B,H,W]."""
    vol = cost_volume.contiguous()
    B, D, H, W = vol.shape
    disp = torch.empty((B, H, W), device=vol.device, dtype=torch.float32)
    with torch.cuda.device(vol.device):
        N.call("stereo_soft_argmin_f32", N.ptr(vol), N.ptr(disp), N.ptr(None), B, D, H, W, 1 if match_similarity else 0,
               _st(vol))
    return disp


@soft_argmin.register_fake
def _(cost_volume, match_similarity):
    B, D, H, W = cost_volume.shape
    return cost_volume.new_empty((B, H, W))


@torch.library.custom_op("stereo_hip::soft_argmin_bwd", mutates_args=(), device_types="cuda")
def soft_argmin_bwd(cost_volume: torch.Tensor, grad_disp: torch.Tensor, match_similarity: bool) -> torch.Tensor:
    vol, gd = cost_volume.contiguous(), grad_disp.contiguous()
    B, D, H, W = vol.shape
    gvol = torch.empty_like(vol)
    with torch.cuda.device(vol.device):
        N.call("stereo_soft_argmin_bwd_f32", N.ptr(vol), N.ptr(gd), N.ptr(None), N.ptr(gvol), B, D, H, W,
               1 if match_similarity else 0, _st(vol))
    return gvol


@soft_argmin_bwd.register_fake
def _(cost_volume, grad_disp, match_similarity):
    return torch.empty_like(cost_volume)


def _sa_setup(ctx, inputs, output):
    vol, ctx.sim = inputs
    ctx.save_for_backward(vol)


soft_argmin.register_autograd(lambda ctx, g: (soft_argmin_bwd(ctx.saved_tensors[0], g, ctx.sim), None), setup_context=_sa_setup)


# ---------------------------------------------------------------------------------------------- fused cfg-2 volume stage
@torch.library.custom_op("stereo_hip::gwc_volume_soft_argmin", mutates_args=(), device_types="cuda")
def gwc_volume_soft_argmin(left: torch.Tensor, right: torch.Tensor, max_disp: int, num_groups: int,
                           match_similarity: bool) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """A3 + group mean + A4 in one launch -> (vol [B,G,D,H,W], disp [B,H,W], mean [B,D,H,W]).  Inference op (the training
    path is ``build_gwc_volume_with_disparity`` outside a compiler, which has its own fused backward)."""
    L, R = left.contiguous(), right.contiguous()
    B, C, H, W = L.shape
    vol = torch.empty((B, num_groups, max_disp, H, W), device=L.device, dtype=torch.float32)
    disp = torch.empty((B, H, W), device=L.device, dtype=torch.float32)
    mean = torch.empty((B, max_disp, H, W), device=L.device, dtype=torch.float32)
    with torch.cuda.device(L.device):
        N.feat_call("stereo_gwc_volume_soft_argmin", [L, R],
                    lambda f: (N.ptr(f[0]), N.ptr(f[1]), N.ptr(vol), N.ptr(mean), N.ptr(disp), B, C, H, W, int(max_disp),
                               int(num_groups), 1 if match_similarity else 0, _st(L)))
    return vol, disp, mean


@gwc_volume_soft_argmin.register_fake
def _(left, right, max_disp, num_groups, match_similarity):
    B, C, H, W = left.shape
    f = dict(dtype=torch.float32)
    return (left.new_empty((B, num_groups, max_disp, H, W), **f), left.new_empty((B, H, W), **f),
            left.new_empty((B, max_disp, H, W), **f))


# ---------------------------------------------------------------------------------------------- A5 local lookup
@torch.library.custom_op("stereo_hip::local_lookup", mutates_args=(), device_types="cuda")
def local_lookup(cost_volume: torch.Tensor, disp: torch.Tensor, radius: float, sample_points: int,
                 consider_valid: bool) -> torch.Tensor:
    """A5: vol [B,D,H,W], disp [B,1,H,W] -> [B,1,H,W]."""
    vol, d = cost_volume.contiguous(), disp.contiguous()
    B, D, H, W = vol.shape
    out = torch.empty((B, 1, H, W), device=vol.device, dtype=torch.float32)
    with torch.cuda.device(vol.device):
        N.call("stereo_local_lookup_f32", N.ptr(vol), N.ptr(d), N.ptr(out), B, D, H, W, float(radius), int(sample_points),
               1 if consider_valid else 0, _st(vol))
    return out


@local_lookup.register_fake
def _(cost_volume, disp, radius, sample_points, consider_valid):
    return torch.empty_like(disp)


@torch.library.custom_op("stereo_hip::local_lookup_bwd", mutates_args=(), device_types="cuda")
def local_lookup_bwd(cost_volume: torch.Tensor, disp: torch.Tensor, grad_out: torch.Tensor, radius: float,
                     sample_points: int, consider_valid: bool) -> tuple[torch.Tensor, torch.Tensor]:
    vol, d, g = cost_volume.contiguous(), disp.contiguous(), grad_out.contiguous()
    B, D, H, W = vol.shape
    gvol, gdisp = torch.zeros_like(vol), torch.empty_like(d)
    with torch.cuda.device(vol.device):
        N.call("stereo_local_lookup_bwd_f32", N.ptr(vol), N.ptr(d), N.ptr(g), N.ptr(gvol), N.ptr(gdisp), B, D, H, W,
               float(radius), int(sample_points), 1 if consider_valid else 0, _st(vol))
    return gvol, gdisp


@local_lookup_bwd.register_fake
def _(cost_volume, disp, grad_out, radius, sample_points, consider_valid):
    return torch.empty_like(cost_volume), torch.empty_like(disp)


def _ll_setup(ctx, inputs, output):
    vol, disp, ctx.radius, ctx.S, ctx.cv = inputs
    ctx.save_for_backward(vol, disp)


def _ll_backward(ctx, grad):
    vol, disp = ctx.saved_tensors
    gv, gd = local_lookup_bwd(vol, disp, grad, ctx.radius, ctx.S, ctx.cv)
    return gv, gd, None, None, None


local_lookup.register_autograd(_ll_backward, setup_context=_ll_setup)


# ---------------------------------------------------------------------------------------------- A7 cross-attention
@torch.library.custom_op("stereo_hip::cross_attention", mutates_args=(), device_types="cuda")
def cross_attention(left: torch.Tensor, right: torch.Tensor, value: torch.Tensor, max_disp: int, mask_mode: int,
                    scale_mode: int) -> tuple[torch.Tensor, torch.Tensor]:
    """A7 (inference op): -> (out [B,Cv,H,W], disp [B,H,W]); mask_mode 0 'zero' / 1 'neginf', scale_mode 0 'sqrt_c' / 1 'mean'."""
    L, R, V = left.contiguous(), right.contiguous(), value.contiguous()
    B, C, H, W = L.shape
    Cv = V.shape[1]
    out = torch.empty((B, Cv, H, W), device=L.device, dtype=torch.float32)
    disp = torch.empty((B, H, W), device=L.device, dtype=torch.float32)
    same = V.data_ptr() == R.data_ptr() and V.shape == R.shape
    with torch.cuda.device(L.device):
        N.feat_call("stereo_cross_attention", [L, R, None if same else V],
                    lambda f: (N.ptr(f[0]), N.ptr(f[1]), N.ptr(f[1] if same else f[2]), N.ptr(out), N.ptr(disp), B, C, Cv, H, W,
                               int(max_disp), int(mask_mode), int(scale_mode), _st(L)))
    return out, disp


@cross_attention.register_fake
def _(left, right, value, max_disp, mask_mode, scale_mode):
    B, C, H, W = left.shape
    return left.new_empty((B, value.shape[1], H, W), dtype=torch.float32), left.new_empty((B, H, W), dtype=torch.float32)


# ---------------------------------------------------------------------------------------------- A13 convex upsampling
@torch.library.custom_op("stereo_hip::upsample_convex", mutates_args=(), device_types="cuda")
def upsample_convex(disp: torch.Tensor, mask: torch.Tensor, factor: int) -> torch.Tensor:
    """A13 (inference op): disp [N,Cd,H,W], mask [N,9 f^2,H,W] -> [N,Cd,fH,fW]."""
    d, m = disp.contiguous(), mask.contiguous()
    n, cd, h, w = d.shape
    out = torch.empty((n, cd, factor * h, factor * w), device=d.device, dtype=torch.float32)
    with torch.cuda.device(d.device):
        N.call("stereo_upsample_convex_f32", N.ptr(d), N.ptr(m), N.ptr(out), n, cd, h, w, int(factor), _st(d))
    return out


@upsample_convex.register_fake
def _(disp, mask, factor):
    n, cd, h, w = disp.shape
    return disp.new_empty((n, cd, factor * h, factor * w))


def tracing() -> bool:
    """True while torch.compile / torch.export is tracing: the module surfaces then emit the registered ops."""
    return torch.compiler.is_compiling()
