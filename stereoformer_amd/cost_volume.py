"""Cost-volume builders behind the reference's module surfaces (A1/A2/A3).

Mirrors ``models/utils/build_cost_volume.py:5-54`` (twin ``models/utils/cost.py``),
``models/CrossAttentionCostVolume/GWC_Cost_Volume.py:5-24`` and
``models/utils/submodule.py:30-41`` -- same names, arguments, output layouts
and error behaviour -- with the body replaced by one HIP launch.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _native as N
from . import ops


class _GwcVolumeFn(torch.autograd.Function):
    """Differentiable group-wise / correlation volume (G = 1): HIP forward + HIP backward
    (``stereo_gwc_volume_bwd_f32``)."""

    @staticmethod
    def forward(ctx, left, right, max_disp, num_groups, squeeze_groups):
        vol = _gwc_forward(left, right, max_disp, num_groups)
        ctx.save_for_backward(left, right)
        ctx.cfg = (int(max_disp), int(num_groups), bool(squeeze_groups))
        return vol[:, 0] if squeeze_groups else vol

    @staticmethod
    def backward(ctx, grad_vol):
        left, right = ctx.saved_tensors
        D, G, squeeze = ctx.cfg
        B, C, H, W = left.shape
        gv = N.dev_f32(grad_vol.unsqueeze(1) if squeeze else grad_vol, "gwc backward: grad")
        gl = torch.empty_like(left) if ctx.needs_input_grad[0] else None
        gr = torch.empty_like(right) if ctx.needs_input_grad[1] else None
        with torch.cuda.device(left.device):
            N.call("stereo_gwc_volume_bwd_f32", N.ptr(left), N.ptr(right), N.ptr(gv), N.ptr(gl), N.ptr(gr), B, C, H, W,
                   D, G, N.stream_ptr(left.device))
        return gl, gr, None, None, None


class _GwcWithDisparityFn(torch.autograd.Function):
    """``(vol, disp, mean) = build_gwc_volume_with_disparity(L, R)`` with a backward that follows the fused forward:
    whatever reaches the mean volume (its own cotangent plus the soft-argmin's) goes through ONE correlation-volume
    backward over a [B,D,H,W] gradient; the G-times larger group-wise backward runs only when ``vol`` itself has a
    consumer.  (The composition build_gwc_volume -> group_mean_soft_argmin would materialise and re-read a dense
    [B,G,D,H,W] gradient even when only the disparity is used.)"""

    @staticmethod
    def forward(ctx, left, right, max_disp, num_groups, match_similarity, out_vol=None, out_disp=None, out_mean=None):
        B, C, H, W = left.shape
        D, G = int(max_disp), int(num_groups)
        if out_vol is not None:   # caller-owned (placed) buffers: outputs written in place into history-free aliases
            vol, disp, mean = out_vol, out_disp, out_mean
            ctx.mark_dirty(vol, disp, mean)
        else:
            vol = torch.empty((B, G, D, H, W), device=left.device, dtype=torch.float32)
            disp = torch.empty((B, H, W), device=left.device, dtype=torch.float32)
            mean = torch.empty((B, D, H, W), device=left.device, dtype=torch.float32)
        with torch.cuda.device(left.device):
            N.call("stereo_gwc_volume_soft_argmin_f32", N.ptr(left), N.ptr(right), N.ptr(vol), N.ptr(mean), N.ptr(disp), B, C, H,
                   W, D, G, 1 if match_similarity else 0, N.stream_ptr(left.device))
        ctx.save_for_backward(left, right, mean)
        ctx.cfg = (D, G, bool(match_similarity))
        ctx.set_materialize_grads(False)  # an unused output must not become a 25 GB tensor of zeros
        return vol, disp, mean

    @staticmethod
    def backward(ctx, g_vol, g_disp, g_mean):
        left, right, mean = ctx.saved_tensors
        D, G, sim = ctx.cfg
        B, C, H, W = left.shape
        dev = left.device
        gm = None
        with torch.cuda.device(dev):
            if g_mean is not None:
                g_mean = N.dev_f32(g_mean, "fused volume backward: grad_mean")
            if g_disp is not None:  # soft-argmin backward on the mean volume
                g_disp = N.dev_f32(g_disp, "fused volume backward: grad_disp")
                gm = torch.empty_like(mean)
                fused_add = False
                if g_mean is not None:  # both cotangents of the mean volume: summed inside the soft-argmin backward
                    try:
                        N.call("stereo_soft_argmin_bwd_add_f32", N.ptr(mean), N.ptr(g_disp), N.ptr(g_mean), N.ptr(gm), B, D, H, W,
                               1 if sim else 0, N.stream_ptr(dev))
                        fused_add = True
                    except N.StereoHipError as e:
                        if e.code != N.ERR_UNSUPPORTED:
                            raise
                if not fused_add:
                    N.call("stereo_soft_argmin_bwd_f32", N.ptr(mean), N.ptr(g_disp), N.ptr(None), N.ptr(gm), B, D, H, W,
                           1 if sim else 0, N.stream_ptr(dev))
                    if g_mean is not None:
                        gm = gm + g_mean
            elif g_mean is not None:
                gm = g_mean
            gl = gr = None

            def vol_bwd(gv, groups):
                a = torch.empty_like(left) if ctx.needs_input_grad[0] else None
                b = torch.empty_like(right) if ctx.needs_input_grad[1] else None
                N.call("stereo_gwc_volume_bwd_f32", N.ptr(left), N.ptr(right), N.ptr(gv), N.ptr(a), N.ptr(b), B, C, H, W, D,
                       groups, N.stream_ptr(dev))
                return a, b

            if gm is not None:  # mean == all-channel correlation volume (identity I1): its backward is the G = 1 one
                gl, gr = vol_bwd(gm.contiguous(), 1)
            if g_vol is not None:
                a, b = vol_bwd(N.dev_f32(g_vol, "fused volume backward: grad_vol"), G)
                gl = a if gl is None else (gl + a if a is not None else gl)
                gr = b if gr is None else (gr + b if b is not None else gr)
        return gl, gr, None, None, None, None, None, None


class _ShiftVolumeFn(torch.autograd.Function):
    """Differentiable 'difference' / 'concat' volume (A2): HIP forward + HIP backward (shifted plane sums)."""

    @staticmethod
    def forward(ctx, left, right, max_disp, concat):
        B, C, H, W = left.shape
        D = int(max_disp)
        vol = torch.empty((B, 2 * C if concat else C, D, H, W), device=left.device, dtype=torch.float32)
        with torch.cuda.device(left.device):
            N.call("stereo_concat_volume_f32" if concat else "stereo_diff_volume_f32", N.ptr(left), N.ptr(right), N.ptr(vol),
                   B, C, H, W, D, N.stream_ptr(left.device))
        ctx.cfg = (B, C, H, W, D, bool(concat))
        return vol

    @staticmethod
    def backward(ctx, grad_vol):
        B, C, H, W, D, concat = ctx.cfg
        gv = N.dev_f32(grad_vol, "shift volume backward: grad")
        gl = torch.empty((B, C, H, W), device=gv.device, dtype=torch.float32) if ctx.needs_input_grad[0] else None
        gr = torch.empty((B, C, H, W), device=gv.device, dtype=torch.float32) if ctx.needs_input_grad[1] else None
        with torch.cuda.device(gv.device):
            N.call("stereo_concat_volume_bwd_f32" if concat else "stereo_diff_volume_bwd_f32", N.ptr(gv), N.ptr(gl),
                   N.ptr(gr), B, C, H, W, D, N.stream_ptr(gv.device))
        return gl, gr, None, None


def _gwc_forward(L, R, maxdisp, num_groups):
    B, C, H, W = L.shape
    vol = torch.empty((B, num_groups, maxdisp, H, W), device=L.device, dtype=torch.float32)
    with torch.cuda.device(L.device):
        N.feat_call("stereo_gwc_volume", [L, R], lambda f: (N.ptr(f[0]), N.ptr(f[1]), N.ptr(vol), B, C, H, W, int(maxdisp),
                                                            int(num_groups), N.stream_ptr(L.device)))
    return vol


def _wants_grad(*ts):
    return torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in ts)


def _pair(left, right, who, feat=False):
    """``feat``: the op has an fp16-feature form -- fp16 tensors pass through as they are."""
    check = N.dev_feat if feat else N.dev_f32
    left = check(left, f"{who}: left_feature")
    right = check(right, f"{who}: right_feature")
    if left.dtype != right.dtype:   # mixed precisions: the fp32 path
        left, right = left.float(), right.float()
    if left.dim() != 4 or left.shape != right.shape:
        raise ValueError(f"{who}: expected two [B,C,H,W] tensors of one shape, got {tuple(left.shape)} "
                         f"and {tuple(right.shape)}")
    N.same_device(left, right)
    return left, right


class CostVolume(nn.Module):
    """``CostVolume(max_disp, feature_similarity)`` (build_cost_volume.py:5-54).

    'correlation' -> [B,D,H,W]; 'difference' -> [B,C,D,H,W]; 'concat' ->
    [B,2C,D,H,W]; anything else raises NotImplementedError (:49-50)."""

    def __init__(self, max_disp, feature_similarity="correlation"):
        super().__init__()
        self.max_disp = max_disp
        self.feature_similarity = feature_similarity

    def forward(self, left_feature, right_feature):
        if self.feature_similarity not in ("difference", "concat", "correlation"):
            raise NotImplementedError
        if self.feature_similarity == "correlation" and ops.tracing():   # torch.compile / export: the registered op
            return torch.ops.stereo_hip.gwc_volume(left_feature, right_feature, int(self.max_disp), 1)[:, 0]
        if self.feature_similarity == "correlation" and _wants_grad(left_feature, right_feature):
            with torch.no_grad():
                _pair(left_feature, right_feature, "CostVolume")  # validation only
            return _GwcVolumeFn.apply(left_feature.contiguous(), right_feature.contiguous(), int(self.max_disp), 1, True)
        if _wants_grad(left_feature, right_feature):   # 'difference' / 'concat' under grad
            with torch.no_grad():
                _pair(left_feature, right_feature, "CostVolume")  # validation only
            return _ShiftVolumeFn.apply(left_feature.contiguous(), right_feature.contiguous(), int(self.max_disp),
                                        self.feature_similarity == "concat")
        L, R = _pair(left_feature, right_feature, "CostVolume", feat=self.feature_similarity == "correlation")
        b, c, h, w = L.shape
        D = int(self.max_disp)
        st = N.stream_ptr(L.device)
        with torch.cuda.device(L.device):
            if self.feature_similarity == "correlation":
                vol = torch.empty((b, D, h, w), device=L.device, dtype=torch.float32)
                N.feat_call("stereo_corr_volume", [L, R], lambda f: (N.ptr(f[0]), N.ptr(f[1]), N.ptr(vol), b, c, h, w, D, st))
            elif self.feature_similarity == "difference":
                vol = torch.empty((b, c, D, h, w), device=L.device, dtype=torch.float32)
                N.call("stereo_diff_volume_f32", N.ptr(L), N.ptr(R), N.ptr(vol), b, c, h, w, D, st)
            else:
                vol = torch.empty((b, 2 * c, D, h, w), device=L.device, dtype=torch.float32)
                N.call("stereo_concat_volume_f32", N.ptr(L), N.ptr(R), N.ptr(vol), b, c, h, w, D, st)
        return vol


class CostVolumePyramid(nn.Module):
    """``CostVolumePyramid`` (models/utils/cost.py:56-72): scale s uses max_disp // 2**s."""

    def __init__(self, max_disp, feature_similarity="correlation"):
        super().__init__()
        self.max_disp = max_disp
        self.feature_similarity = feature_similarity

    def forward(self, left_feature_pyramid, right_feature_pyramid):
        out = []
        for s in range(len(left_feature_pyramid)):
            cv = CostVolume(self.max_disp // (2 ** s), self.feature_similarity)
            out.append(cv(left_feature_pyramid[s], right_feature_pyramid[s]))
        return out


def build_gwc_volume(refimg_fea, targetimg_fea, maxdisp, num_groups):
    """``build_gwc_volume`` (GWC_Cost_Volume.py:14-24) -> [B,G,D,H,W]."""
    if ops.tracing():
        return torch.ops.stereo_hip.gwc_volume(refimg_fea, targetimg_fea, int(maxdisp), int(num_groups))
    if _wants_grad(refimg_fea, targetimg_fea):
        with torch.no_grad():
            L, R = _pair(refimg_fea, targetimg_fea, "build_gwc_volume")
        assert L.shape[1] % num_groups == 0  # GWC_Cost_Volume.py:7
        return _GwcVolumeFn.apply(refimg_fea.contiguous(), targetimg_fea.contiguous(), int(maxdisp), int(num_groups), False)
    L, R = _pair(refimg_fea, targetimg_fea, "build_gwc_volume", feat=True)
    assert L.shape[1] % num_groups == 0  # GWC_Cost_Volume.py:7
    return _gwc_forward(L, R, maxdisp, num_groups)


def corr_volume_with_disparity(left_feature, right_feature, max_disp, match_similarity=True):
    """``vol = CostVolume(D, 'correlation')(L, R)`` and ``DisparityEstimation(D, match_similarity)(vol)`` from ONE
    kernel: each finished disparity strip is stored and folded into the online softmax in the same pass, so the
    volume is never read back -> ``(vol [B,D,H,W], disp [B,H,W])``.  Differentiable (composition in grad mode)."""
    from .estimation import DisparityEstimation
    if _wants_grad(left_feature, right_feature):
        vol = CostVolume(int(max_disp), "correlation")(left_feature, right_feature)
        return vol, DisparityEstimation(int(max_disp), match_similarity)(vol)
    L, R = _pair(left_feature, right_feature, "corr_volume_with_disparity", feat=True)
    B, C, H, W = L.shape
    D = int(max_disp)
    vol = torch.empty((B, D, H, W), device=L.device, dtype=torch.float32)
    disp = torch.empty((B, H, W), device=L.device, dtype=torch.float32)
    with torch.cuda.device(L.device):
        N.feat_call("stereo_corr_volume_soft_argmin", [L, R],
                    lambda f: (N.ptr(f[0]), N.ptr(f[1]), N.ptr(vol), N.ptr(disp), B, C, H, W, D,
                               1 if match_similarity else 0, N.stream_ptr(L.device)))
    return vol, disp


def _check_out(t, shape, dev, what):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.device == dev and t.dtype == torch.float32
            and tuple(t.shape) == tuple(shape) and t.is_contiguous()):
        raise ValueError(f"build_gwc_volume_with_disparity: out {what} must be a contiguous float32 tensor of shape "
                         f"{tuple(shape)} on {dev}")
    return t


def build_gwc_volume_with_disparity(refimg_fea, targetimg_fea, maxdisp, num_groups, return_mean=False,
                                    match_similarity=True, out=None):
    """``vol = build_gwc_volume(L, R, D, G)`` together with ``DisparityEstimation(D)(vol.mean(1))`` (and the mean
    volume itself when ``return_mean``) in one call -> ``(vol, disp[, mean])``.

    ``vol.mean(1)`` over equally sized groups is the all-channel correlation volume (SURVEY 8(c) identity I1), so
    the disparity and the mean come from the LDS-staged feature tiles while the group-wise volume is being
    produced: the [B,G,D,H,W] volume is written once and never read back (cfg-2: 25.5 GB of reads saved).
    Differentiable: the backward follows the fused forward (see ``_GwcWithDisparityFn``).

    ``out=(vol, disp, mean)`` (like torch's ``out=``): caller-owned output tensors that this call overwrites completely -- for
    buffers that live across many steps and were placed with ``stereoformer_amd.hbm.place_volume_stage_outputs``.  Under grad
    the outputs are in-place results on history-free aliases of the buffers (a training loop whose backward runs before the
    next forward can reuse the same placed buffers every step)."""
    if ops.tracing():
        vol, disp, mean = torch.ops.stereo_hip.gwc_volume_soft_argmin(refimg_fea, targetimg_fea, int(maxdisp), int(num_groups),
                                                                      bool(match_similarity))
        return (vol, disp, mean) if return_mean else (vol, disp)
    if _wants_grad(refimg_fea, targetimg_fea):
        with torch.no_grad():
            L, R = _pair(refimg_fea, targetimg_fea, "build_gwc_volume_with_disparity")
        assert L.shape[1] % num_groups == 0  # GWC_Cost_Volume.py:7
        outs = ()
        if out is not None:
            # under grad the buffers are written in place through history-free aliases (`detach()`): the returned tensors
            # carry this call's graph, the caller's buffers stay plain tensors for the next step.  As with any in-place op,
            # a backward through an EARLIER call's graph after the buffers were overwritten is refused by autograd.
            B, C, H, W = L.shape
            D, G = int(maxdisp), int(num_groups)
            if any(isinstance(t, torch.Tensor) and t.requires_grad for t in out):
                raise RuntimeError("build_gwc_volume_with_disparity: out= tensors must not require grad")
            outs = (_check_out(out[0], (B, G, D, H, W), L.device, "[0] (volume)").detach(),
                    _check_out(out[1], (B, H, W), L.device, "[1] (disparity)").detach(),
                    _check_out(out[2], (B, D, H, W), L.device, "[2] (mean volume)").detach())
        vol, disp, mean = _GwcWithDisparityFn.apply(refimg_fea.contiguous(), targetimg_fea.contiguous(), int(maxdisp),
                                                    int(num_groups), bool(match_similarity), *outs)
        return (vol, disp, mean) if return_mean else (vol, disp)
    L, R = _pair(refimg_fea, targetimg_fea, "build_gwc_volume_with_disparity", feat=True)
    assert L.shape[1] % num_groups == 0  # GWC_Cost_Volume.py:7
    B, C, H, W = L.shape
    D, G = int(maxdisp), int(num_groups)
    if out is not None:
        vol = _check_out(out[0], (B, G, D, H, W), L.device, "[0] (volume)")
        disp = _check_out(out[1], (B, H, W), L.device, "[1] (disparity)")
        mean = _check_out(out[2], (B, D, H, W), L.device, "[2] (mean volume)")
    else:
        vol = torch.empty((B, G, D, H, W), device=L.device, dtype=torch.float32)
        disp = torch.empty((B, H, W), device=L.device, dtype=torch.float32)
        # the fp16 form always produces the mean volume (the fused kernel's only shape); fp32 keeps it optional
        mean = torch.empty((B, D, H, W), device=L.device, dtype=torch.float32) if (return_mean or L.dtype == torch.float16) else None
    with torch.cuda.device(L.device):
        N.feat_call("stereo_gwc_volume_soft_argmin", [L, R],
                    lambda f: (N.ptr(f[0]), N.ptr(f[1]), N.ptr(vol), N.ptr(mean), N.ptr(disp), B, C, H, W, D, G,
                               1 if match_similarity else 0, N.stream_ptr(L.device)))
    return (vol, disp, mean) if return_mean else (vol, disp)


def groupwise_correlation(fea1, fea2, num_groups):
    """``groupwise_correlation`` (GWC_Cost_Volume.py:5-11) -> [B,G,H,W] (= the D=1 slice)."""
    L, R = _pair(fea1, fea2, "groupwise_correlation", feat=True)
    B, C, H, W = L.shape
    assert C % num_groups == 0
    cost = build_gwc_volume(L, R, 1, num_groups).view(B, num_groups, H, W)
    assert cost.shape == (B, num_groups, H, W)
    return cost


def build_concat_volume(refimg_fea, targetimg_fea, maxdisp):
    """``build_concat_volume`` (models/utils/submodule.py:30-41) == CostVolume(maxdisp,'concat')."""
    return CostVolume(maxdisp, "concat")(refimg_fea, targetimg_fea)
