"""GRU-iterated local cost-volume updater (A11) behind
``models/LocalCostVolume/Attempts/update.py`` (and ``updatev2.py``).

The 3x3 convolutions / BatchNorm stay with the host framework (MIOpen); everything
else on the iteration path is a HIP kernel: image resize, ``disp_warp``, the uncertainty
volume (softmax * d * (d-disp)^2 in one pass), the GRU gate math, and the learned-bounds
sampler.  Parameter names (including the reference's aliased ``conv_zz.0`` etc.) match the
reference so its ``state_dict``s load unchanged.  No host sync inside ``forward``.  Trainable: every custom op on
the path has a HIP backward (call under ``torch.no_grad()`` for inference).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _native as N
from .disparity_warper import disp_warp
from .local_cost_volume import bounds_lookup, uncertainty_volume
from .upsample import resize_bilinear


def _wants_grad(*ts):
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in ts)


class _GruResetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, rc, cr):
        ctx.save_for_backward(h, rc, cr if cr is not None else h.new_empty(0))
        ctx.has_cr = cr is not None
        return gru_reset(h, rc, cr)

    @staticmethod
    def backward(ctx, g):
        h, rc, cr = ctx.saved_tensors
        cr = cr if ctx.has_cr else None
        g = N.dev_f32(g, "gru_reset backward: grad")
        gh, grc = torch.empty_like(h), torch.empty_like(h)
        with torch.cuda.device(h.device):
            N.call("stereo_gru_reset_bwd_f32", N.ptr(h), N.ptr(rc), N.ptr(cr), N.ptr(g), N.ptr(gh), N.ptr(grc), h.numel(),
                   N.stream_ptr(h.device))
        return gh, grc, (grc if ctx.has_cr else None)


class _GruBlendFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, zc, qc, cz, cq, softsign):
        e = h.new_empty(0)
        ctx.save_for_backward(h, zc, qc, cz if cz is not None else e, cq if cq is not None else e)
        ctx.flags = (cz is not None, cq is not None, bool(softsign))
        return gru_blend(h, zc, qc, cz, cq, softsign)

    @staticmethod
    def backward(ctx, g):
        h, zc, qc, cz, cq = ctx.saved_tensors
        has_cz, has_cq, softsign = ctx.flags
        g = N.dev_f32(g, "gru_blend backward: grad")
        gh, gz, gq = torch.empty_like(h), torch.empty_like(h), torch.empty_like(h)
        with torch.cuda.device(h.device):
            N.call("stereo_gru_blend_bwd_f32", N.ptr(h), N.ptr(zc), N.ptr(qc), N.ptr(cz if has_cz else None),
                   N.ptr(cq if has_cq else None), N.ptr(g), N.ptr(gh), N.ptr(gz), N.ptr(gq), h.numel(),
                   1 if softsign else 0, N.stream_ptr(h.device))
        return gh, gz, gq, (gz if has_cz else None), (gq if has_cq else None), None


def _all_f16(*ts):
    live = [t for t in ts if t is not None]
    return all(t.dtype == torch.float16 and t.is_cuda for t in live)


def _dense16(t):
    """fp16 operand of a pointwise op: NCHW- or channels-last-dense."""
    if t is None or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last)):
        return t
    return t.contiguous()


def gru_reset(h, rc, cr=None):
    """``sigmoid(rc + cr) * h`` in one kernel (differentiable: HIP backward).  fp32 tensors; all-fp16 operands (the
    update block under ``mixed_precision``) run the fp16-in / fp16-out kernel (inference only)."""
    if _all_f16(h, rc, cr) and not _wants_grad(h, rc, cr):
        h, rc, cr = N.one_layout(_dense16(h), _dense16(rc), _dense16(cr))
        if rc.shape != h.shape or (cr is not None and cr.shape != h.shape):
            raise ValueError("gru_reset: shape mismatch")
        out = torch.empty_like(h)
        with torch.cuda.device(h.device):
            N.call("stereo_gru_reset_f16", N.ptr(h), N.ptr(rc), N.ptr(cr), N.ptr(out), h.numel(), N.stream_ptr(h.device))
        return out
    if _wants_grad(h, rc, cr):
        return _GruResetFn.apply(h.contiguous(), rc.contiguous(), None if cr is None else cr.contiguous())
    h = N.dev_f32(h, "gru_reset: h", dense_ok=True)
    rc = N.dev_f32(rc, "gru_reset: rc", dense_ok=True)
    cr = None if cr is None else N.dev_f32(cr, "gru_reset: cr", dense_ok=True)
    if rc.shape != h.shape or (cr is not None and cr.shape != h.shape):
        raise ValueError("gru_reset: shape mismatch")
    h, rc, cr = N.one_layout(h, rc, cr)   # NCHW or channels-last, but one layout for all (flat indexing)
    out = torch.empty_like(h)
    with torch.cuda.device(h.device):
        N.call("stereo_gru_reset_f32", N.ptr(h), N.ptr(rc), N.ptr(cr), N.ptr(out), h.numel(), N.stream_ptr(h.device))
    return out


def gru_blend(h, zc, qc, cz=None, cq=None, softsign=False):
    """``(1-z)*h + z*act(qc+cq)`` with ``z = sigmoid(zc+cz)`` in one kernel (differentiable: HIP backward).  fp32
    tensors; all-fp16 operands run the fp16-in / fp16-out kernel (inference only)."""
    if _all_f16(h, zc, qc, cz, cq) and not _wants_grad(h, zc, qc, cz, cq):
        h, zc, qc, cz, cq = N.one_layout(_dense16(h), _dense16(zc), _dense16(qc), _dense16(cz), _dense16(cq))
        for t in (zc, qc, cz, cq):
            if t is not None and t.shape != h.shape:
                raise ValueError("gru_blend: shape mismatch")
        out = torch.empty_like(h)
        with torch.cuda.device(h.device):
            N.call("stereo_gru_blend_f16", N.ptr(h), N.ptr(zc), N.ptr(qc), N.ptr(cz), N.ptr(cq), N.ptr(out), h.numel(),
                   1 if softsign else 0, N.stream_ptr(h.device))
        return out
    if _wants_grad(h, zc, qc, cz, cq):
        c = lambda t: None if t is None else t.contiguous()
        return _GruBlendFn.apply(h.contiguous(), zc.contiguous(), qc.contiguous(), c(cz), c(cq), bool(softsign))
    h = N.dev_f32(h, "gru_blend: h", dense_ok=True)
    zc = N.dev_f32(zc, "gru_blend: zc", dense_ok=True)
    qc = N.dev_f32(qc, "gru_blend: qc", dense_ok=True)
    cz = None if cz is None else N.dev_f32(cz, "gru_blend: cz", dense_ok=True)
    cq = None if cq is None else N.dev_f32(cq, "gru_blend: cq", dense_ok=True)
    for t in (zc, qc, cz, cq):
        if t is not None and t.shape != h.shape:
            raise ValueError("gru_blend: shape mismatch")
    h, zc, qc, cz, cq = N.one_layout(h, zc, qc, cz, cq)
    out = torch.empty_like(h)
    with torch.cuda.device(h.device):
        N.call("stereo_gru_blend_f32", N.ptr(h), N.ptr(zc), N.ptr(qc), N.ptr(cz), N.ptr(cq), N.ptr(out), h.numel(),
               1 if softsign else 0, N.stream_ptr(h.device))
    return out


def guidance_inputs_nhwc(old_cost_volume, cur_disp, left, right, err_channels=16, unc_channels=32):
    """Inputs of ``BasicGuidanceNet``'s two encoders in one launch (``stereo_guidance_inputs_nhwc_f32``), channels-last and
    zero-padded to multiples of 16 channels: ``(disp_warp(right, disp) - left, uncertainty_volume(vol, disp))`` as
    ``[B,err_channels,h,w]`` / ``[B,unc_channels,h,w]`` ``torch.channels_last`` tensors.  Inference only."""
    vol = N.dev_f32(old_cost_volume, "guidance_inputs_nhwc: cost volume")
    disp = N.dev_f32(cur_disp, "guidance_inputs_nhwc: disparity")
    left, right = N.dev_f32(left, "guidance_inputs_nhwc: left"), N.dev_f32(right, "guidance_inputs_nhwc: right")
    B, D, H, W = vol.shape
    Ci = left.shape[1]
    if disp.shape != (B, 1, H, W) or left.shape != (B, Ci, H, W) or right.shape != left.shape:
        raise ValueError("guidance_inputs_nhwc: shape mismatch")
    ce, cu = max(err_channels, (Ci + 15) // 16 * 16), max(unc_channels, (D + 15) // 16 * 16)
    cl = torch.channels_last
    err = torch.empty((B, ce, H, W), device=vol.device, dtype=torch.float32, memory_format=cl)
    unc = torch.empty((B, cu, H, W), device=vol.device, dtype=torch.float32, memory_format=cl)
    with torch.cuda.device(vol.device):
        N.call("stereo_guidance_inputs_nhwc_f32", N.ptr(vol), N.ptr(disp), N.ptr(left), N.ptr(right), N.ptr(err), N.ptr(unc), B, D, Ci,
               H, W, ce, cu, N.stream_ptr(vol.device))
    return err, unc


def _split_convs_ok(*ts):
    """Inference on channels-last fp32 HIP activations outside autocast / tracing: where the 3x3 convolutions go to
    ``conv_split`` (see ``update_raft._split_ok``)."""
    if torch.is_grad_enabled() or torch.is_autocast_enabled("cuda") or torch.compiler.is_compiling():
        return False
    return all(t.is_cuda and t.dtype == torch.float32 and t.dim() == 4 for t in ts) and \
        ts[0].is_contiguous(memory_format=torch.channels_last) and not ts[0].is_contiguous()


class ConvGRU(nn.Module):
    """``ConvGRU(nb_channel, softsign)`` (Attempts/update.py:17-49)."""

    def __init__(self, nb_channel, softsign):
        super().__init__()
        self.conv_z = nn.Conv2d(nb_channel * 2, nb_channel, 3, 1, 1)
        self.conv_b = nn.Conv2d(nb_channel * 2, nb_channel, 3, 1, 1)
        self.conv_g = nn.Conv2d(nb_channel * 2, nb_channel, 3, 1, 1)
        nn.init.orthogonal_(self.conv_z.weight)
        nn.init.orthogonal_(self.conv_b.weight)
        nn.init.orthogonal_(self.conv_g.weight)
        nn.init.constant_(self.conv_z.bias, 0.)
        nn.init.constant_(self.conv_b.bias, 0.)
        nn.init.constant_(self.conv_g.bias, 0.)
        # same aliased containers as the reference so state_dict keys line up (:29-34)
        self.conv_zz = nn.Sequential(self.conv_z, nn.Sigmoid())
        self.conv_bb = nn.Sequential(self.conv_b, nn.Sigmoid())
        self.conv_gg = nn.Sequential(self.conv_g, nn.Softsign() if softsign else nn.Tanh())
        self.softsign = bool(softsign)
        self.nb_channel = nb_channel

    def forward(self, input, prev_h):
        """``input``: the GRU input tensor, or a list of channels-last tensors whose concatenation it is (the guidance
        encoder's two feature maps: never concatenated on the ``conv_split`` path)."""
        parts = list(input) if isinstance(input, (list, tuple)) else [input]
        x0 = parts[0]
        if prev_h is None:  # the reference hard-codes .cuda() here (:38-39)
            prev_h = torch.zeros(x0.size(0), self.nb_channel, x0.size(2), x0.size(3), device=x0.device, dtype=x0.dtype)
        if _split_convs_ok(*parts, prev_h):
            from . import conv_split as C
            if prev_h.is_contiguous() and not prev_h.is_contiguous(memory_format=torch.channels_last):
                prev_h = prev_h.contiguous(memory_format=torch.channels_last)
            srcs = [*parts, prev_h]                                   # the reference's cat order (:40)
            if (prev_h.is_contiguous(memory_format=torch.channels_last) and len(srcs) <= 4 and C.supported(srcs, [self.conv_z])
                    and C.supported(srcs, [self.conv_b]) and C.supported([prev_h, *parts], [self.conv_g])):
                # inference in torch.channels_last: the three gate convolutions on conv_split (no concatenation is
                # materialised) with the gate math in their epilogues (:41-48): two launches per GRU step
                if C.supported(srcs, [self.conv_z, self.conv_b]):
                    z, s = C.conv_nhwc(srcs, [self.conv_z, self.conv_b], epilogue=[("sigmoid", None), ("sigmoid_mul", None, prev_h)])
                else:
                    (z,) = C.conv_nhwc(srcs, [self.conv_z], epilogue=[("sigmoid", None)])
                    (s,) = C.conv_nhwc(srcs, [self.conv_b], epilogue=[("sigmoid_mul", None, prev_h)])
                blend = "gru_blend_softsign" if self.softsign else "gru_blend"
                return C.conv_nhwc([s, *parts], [self.conv_g], epilogue=[(blend, None, prev_h, z)])[0]
        input = parts[0] if len(parts) == 1 else torch.cat(parts, 1)
        x1 = torch.cat((input, prev_h), 1)
        zc = self.conv_z(x1)
        bc = self.conv_b(x1)
        s = gru_reset(prev_h, bc)                       # b * prev_h
        gc = self.conv_g(torch.cat((s, input), 1))
        return gru_blend(prev_h, zc, gc, softsign=self.softsign)


class BasicGuidanceNet(nn.Module):
    """``BasicGuidanceNet`` (Attempts/update.py:52-92)."""

    def __init__(self, input_channels, hidden_layer=32):
        super().__init__()
        self.input_channels = input_channels
        self.hidden_layers = hidden_layer
        self.uncertain_encoder = nn.Sequential(
            nn.Conv2d(self.input_channels, self.hidden_layers, kernel_size=3, stride=1, padding=1, bias=False),
            nn.BatchNorm2d(self.hidden_layers), nn.ReLU(inplace=True))
        self.disparity_error_encoder = nn.Sequential(
            nn.Conv2d(3, self.hidden_layers, kernel_size=3, stride=1, padding=1, bias=False),
            nn.BatchNorm2d(self.hidden_layers), nn.ReLU(inplace=True))

    def features(self, old_cost_volume, cur_disp, left_image, right_image, nhwc=False):
        """``[error_feature, uncertainity_feature]`` (:75-92, un-concatenated).  ``nhwc`` (inference, eval-mode BatchNorm, fp32 HIP
        tensors): the whole encoder is three launches -- ``stereo_guidance_inputs_nhwc_f32`` (warp + error map + uncertainty
        volume, channels-last, zero-padded to 16 / 32 channels) and one ``conv_split`` launch per branch with the BatchNorm
        folded into the packed weights and the ReLU in the epilogue -- instead of resize x 2, warp, subtract, uncertainty, two
        library convolutions, two BatchNorm and two ReLU launches."""
        if cur_disp.size(-1) != left_image.size(-1):
            size = (cur_disp.size(-2), cur_disp.size(-1))
            cur_left_image = resize_bilinear(left_image, size, align_corners=False)
            cur_right_image = resize_bilinear(right_image, size, align_corners=False)
        else:
            cur_left_image, cur_right_image = left_image, right_image
        ce, be = self.disparity_error_encoder[0], self.disparity_error_encoder[1]
        cu, bu = self.uncertain_encoder[0], self.uncertain_encoder[1]
        if (nhwc and not torch.is_grad_enabled() and not torch.is_autocast_enabled("cuda") and not torch.compiler.is_compiling()
                and not be.training and not bu.training and old_cost_volume.is_cuda and old_cost_volume.dtype == torch.float32):
            from . import conv_split as C
            err, unc = guidance_inputs_nhwc(old_cost_volume, cur_disp, cur_left_image, cur_right_image)
            if C.supported([err], [ce], pad_in=True) and C.supported([unc], [cu], pad_in=True):
                (ef,) = C.conv_nhwc([err], [ce], relu=True, bns=[be], pad_in=True)
                (uf,) = C.conv_nhwc([unc], [cu], relu=True, bns=[bu], pad_in=True)
                return [ef, uf]
        warped_left, _ = disp_warp(cur_right_image, cur_disp)
        error_map = warped_left - cur_left_image
        uncertainity_volume = uncertainty_volume(old_cost_volume, cur_disp)
        error_feature = self.disparity_error_encoder(error_map)
        uncertainity_feature = self.uncertain_encoder(uncertainity_volume)
        return [error_feature, uncertainity_feature]

    def forward(self, old_cost_volume, cur_disp, left_image, right_image):
        return torch.cat(self.features(old_cost_volume, cur_disp, left_image, right_image), dim=1)


class OffsetPredictionHead(nn.Module):
    """``OffsetPredictionHead`` (Attempts/update.py:96-105): two non-negative offsets."""

    def __init__(self, input_dim, hidden_dim):
        super().__init__()
        self.conv1 = nn.Conv2d(input_dim, hidden_dim, 3, padding=1)
        self.conv2 = nn.Conv2d(hidden_dim, 2, 3, padding=1)
        self.relu1 = nn.ReLU(inplace=True)
        self.relu2 = nn.ReLU(inplace=True)

    def forward(self, x):
        if _split_convs_ok(x):
            from . import conv_split as C
            if C.supported([x], [self.conv1]):
                (y,) = C.conv_nhwc([x], [self.conv1], relu=True)
                if C.small_supported(y, self.conv2):
                    return C.conv3x3_small(y, self.conv2, relu=True)      # (64 -> 2: fp32 vector kernel, ReLU inside, NCHW out)
                return self.relu2(self.conv2(y))
        return self.relu2(self.conv2(self.relu1(self.conv1(x))))


class DisparityUpdateDLC(nn.Module):
    """``DisparityUpdateDLC`` (Attempts/update.py:178-242, the definition that survives):
    as ...WithMask but without the mask head; its offset head takes 32 channels, i.e. it
    only works with hidden_dim == 16 (a reference quirk kept as is)."""

    def __init__(self, input_channels, hidden_dim, sample_points=10):
        super().__init__()
        self.sample_points = sample_points
        self.hidden_dim = hidden_dim
        self.encoder = BasicGuidanceNet(input_channels=input_channels, hidden_layer=self.hidden_dim)
        self.gru = ConvGRU(nb_channel=self.hidden_dim * 2, softsign=False)
        self.offset = OffsetPredictionHead(input_dim=32, hidden_dim=64)

    def forward(self, old_cost_volume, cur_disp, left_image, right_image, hidden_state=None, consider_valid=True):
        offset_features = self.encoder(old_cost_volume, cur_disp, left_image, right_image)
        hidden_state = self.gru(offset_features, hidden_state)
        bounds = self.offset(hidden_state)
        disp = bounds_lookup(old_cost_volume, cur_disp, bounds, self.sample_points, consider_valid)
        return disp, hidden_state


class DisparityUpdateDLCWithMask(nn.Module):
    """``DisparityUpdateDLCWithMask(input_channels, hidden_dim, sample_points=10)``
    (Attempts/update.py:245-316) -> (disp [B,1,h,w], hidden [B,2*hidden_dim,h,w], mask [B,576,h,w])."""

    def __init__(self, input_channels, hidden_dim, sample_points=10):
        super().__init__()
        self.sample_points = sample_points
        self.hidden_dim = hidden_dim
        self.encoder = BasicGuidanceNet(input_channels=input_channels, hidden_layer=self.hidden_dim)
        self.gru = ConvGRU(nb_channel=self.hidden_dim * 2, softsign=False)
        self.offset = OffsetPredictionHead(input_dim=self.hidden_dim * 2, hidden_dim=64)
        self.mask = nn.Sequential(
            nn.Conv2d(self.hidden_dim * 2, 256, 3, padding=1), nn.ReLU(inplace=True),
            nn.Conv2d(256, 64 * 9, 1, padding=0))

    def forward(self, old_cost_volume, cur_disp, left_image, right_image, hidden_state=None, consider_valid=True,
                need_mask=True):
        """``need_mask=False`` (not in the reference) skips the upsampling-mask head (returns ``None`` for it): an
        inference loop that only upsamples its last iterate does not need the earlier masks."""
        nhwc = hidden_state is not None and _split_convs_ok(hidden_state)   # the loop keeps the hidden state channels-last
        offset_features = self.encoder.features(old_cost_volume, cur_disp, left_image, right_image, nhwc=nhwc)
        hidden_state = self.gru(offset_features, hidden_state)
        mask = None
        if need_mask:
            m0 = None
            if _split_convs_ok(hidden_state):
                from . import conv_split as C
                if C.supported([hidden_state], [self.mask[0]]):
                    (m0,) = C.conv_nhwc([hidden_state], [self.mask[0]], relu=True)
            if m0 is not None and C.supported([m0], [self.mask[2]]):
                # the 1x1 projection on the same kernel, the 0.25 folded into its packed weights (the library's 1x1 + a
                # separate scaling pass were 0.5 ms of the 5.5-ms loop: once per call, but at 576 output channels)
                (mask,) = C.conv_nhwc([m0], [self.mask[2]], scales=[0.25])
            else:
                mask = .25 * (self.mask[2](m0) if m0 is not None else self.mask(hidden_state))
        bounds = self.offset(hidden_state)          # [B,2,h,w]: (lower offset, upper offset)
        disp = bounds_lookup(old_cost_volume, cur_disp, bounds, self.sample_points, consider_valid)
        return disp, hidden_state, mask


class DisparityUpdateDLCWithMaskV2(nn.Module):
    """The ``updatev2.py:169-251`` variant: an extra left-feature encoder (256 -> 64 ch,
    hard-wired in the reference) concatenated into the GRU input."""

    def __init__(self, input_channels, feature_dim, hidden_dim, sample_points=10):
        super().__init__()
        self.sample_points = sample_points
        self.hidden_dim = hidden_dim
        self.feature_dim = feature_dim
        self.encoder = BasicGuidanceNet(input_channels=input_channels, hidden_layer=self.hidden_dim)
        self.feature_encode = nn.Sequential(
            nn.Conv2d(256, 64, kernel_size=3, stride=1, padding=1, bias=False), nn.BatchNorm2d(64),
            nn.ReLU(inplace=True))
        ch = self.hidden_dim * 2 + self.feature_dim
        self.gru = ConvGRU(nb_channel=ch, softsign=False)
        self.offset = OffsetPredictionHead(input_dim=ch, hidden_dim=64)
        self.mask = nn.Sequential(nn.Conv2d(ch, 256, 3, padding=1), nn.ReLU(inplace=True),
                                  nn.Conv2d(256, 64 * 9, 1, padding=0))

    def forward(self, old_cost_volume, cur_disp, left_image, right_image, hidden_state=None, left_feature=None,
                consider_valid=True):
        offset_features = self.encoder(old_cost_volume, cur_disp, left_image, right_image)
        input_features = torch.cat((offset_features, self.feature_encode(left_feature)), dim=1)
        hidden_state = self.gru(input_features, hidden_state)
        mask = .25 * self.mask(hidden_state)
        bounds = self.offset(hidden_state)
        disp = bounds_lookup(old_cost_volume, cur_disp, bounds, self.sample_points, consider_valid)
        return disp, hidden_state, mask
