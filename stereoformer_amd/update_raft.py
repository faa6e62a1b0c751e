"""RAFT-Stereo multi-level ConvGRU update operator (A14).

Host-side mirror of ``models/IterativeLocalCostVolume/update/update.py`` (module and parameter names
are kept so the reference's ``state_dict``s load unchanged).  Convolutions and pooling are left to
the host framework (MIOpen); what this package owns on the path is the gate math -- ``sigmoid(conv +
c) * h`` and ``(1 - z) * h + z * tanh(conv + c)`` as single HIP kernels -- and the bilinear ``interp``; all three
have HIP backward kernels, so the block trains (call under ``torch.no_grad()`` for inference).
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import conv_split
from .update_dlc import gru_blend, gru_reset
from .upsample import interp


def _conv(cin, cout, k, pad=None):
    if isinstance(k, tuple):
        return nn.Conv2d(cin, cout, k, padding=pad)
    return nn.Conv2d(cin, cout, k, padding=k // 2 if pad is None else pad)


def _avg_pool(x, k, stride, pad):
    """``F.avg_pool2d(x, k, stride=stride, padding=pad)``; channels-last fp32 HIP activations outside autograd go to the
    library's own pooling kernel (``stereo_avg_pool_nhwc_f32``)."""
    if (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] % 4 == 0 and not x.is_contiguous()
            and x.is_contiguous(memory_format=torch.channels_last) and not (torch.is_grad_enabled() and x.requires_grad)
            and not torch.compiler.is_compiling()):
        from . import _native as N
        B, C, H, W = x.shape
        ho, wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        out = torch.empty((B, C, ho, wo), device=x.device, dtype=torch.float32, memory_format=torch.channels_last)
        with torch.cuda.device(x.device):
            N.call("stereo_avg_pool_nhwc_f32", N.ptr(x), N.ptr(out), B, C, H, W, k, stride, pad, N.stream_ptr(x.device))
        return out
    return F.avg_pool2d(x, k, stride=stride, padding=pad)


def pool2x(x):
    """3x3 / stride-2 average pooling (update.py:119-120)."""
    return _avg_pool(x, 3, 2, 1)


def pool4x(x):
    """5x5 / stride-4 average pooling (update.py:122-123)."""
    return _avg_pool(x, 5, 4, 1)


def _f32(t):
    """Gate math runs in fp32: under ``autocast`` (``mixed_precision``) the convolutions hand over fp16."""
    return t if t is None or t.dtype == torch.float32 else t.float()


def _fp16_inference():
    """fp16 autocast region without gradients: activations stay fp16 between the convolutions."""
    return (torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.float16
            and not torch.is_grad_enabled())


def _to16(t):
    return None if t is None else (t if t.dtype == torch.float16 else t.half())


def _gru_step(h, x, conv_z, conv_r, conv_q, cz=None, cr=None, cq=None):
    """One ConvGRU update with optional context biases; the two pointwise stages are HIP kernels.  Under fp16
    ``autocast`` without gradients the hidden state and the gates stay fp16 end to end (as in the reference, where the
    autocast region turns ``net`` into fp16 after the first iteration); otherwise the gate math takes fp32 tensors."""
    if _fp16_inference():
        h, x = _to16(h), _to16(x)
        hx = torch.cat((h, x), 1)
        gated = gru_reset(h, conv_r(hx), _to16(cr))
        return gru_blend(h, conv_z(hx), conv_q(torch.cat((gated, x), 1)), _to16(cz), _to16(cq))
    hx = torch.cat((h, x), 1)
    z_pre = conv_z(hx)
    gated = gru_reset(_f32(h), _f32(conv_r(hx)), _f32(cr))
    return gru_blend(_f32(h), _f32(z_pre), _f32(conv_q(torch.cat((gated, x), 1))), _f32(cz), _f32(cq))


def _split_ok(*ts):
    """Inference on channels-last HIP activations -- fp32 outside autocast (the parity configuration: split-bf16 arithmetic), or
    all-fp16 inside an fp16 autocast region (``mixed_precision``: native fp16 MFMA): the envelope of ``conv_split.conv_nhwc``."""
    if torch.is_grad_enabled() or torch.compiler.is_compiling():   # (a traced graph keeps the library convolutions)
        return False
    want = torch.float16 if _fp16_inference() else (None if torch.is_autocast_enabled("cuda") else torch.float32)
    if want is None:
        return False
    return all(t is None or (t.is_cuda and t.dtype == want and t.dim() == 4 and not t.is_contiguous()
                             and t.is_contiguous(memory_format=torch.channels_last)) for t in ts)


_SIDE = {}


def _side_stream(device):
    """One side stream per device for the motion encoder (see ``BasicMultiUpdateBlock_LZH.forward``)."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    st = _SIDE.get(idx)
    if st is None:
        st = _SIDE[idx] = torch.cuda.Stream(device=idx)
    return st


def _gru_step_split(h, xs, conv_z, conv_r, conv_q, cz, cr, cq):
    """The ConvGRU update as TWO launches of ``conv_split`` (inference, channels-last): ``convz | convr`` over the sources
    ``[h | x...]`` (no concatenation is materialised) with the gate math in the epilogue -- ``z = sigmoid(. + cz)`` and
    ``r h = sigmoid(. + cr) * h`` leave the kernel, the pre-activations never exist in memory --, then ``convq`` over
    ``[r h | x...]`` whose epilogue is ``tanh(. + cq)`` and the ``(1 - z) h + z q`` blend (update.py:28-32)."""
    if h.dtype == torch.float16:
        # mixed_precision: an fp16 epilogue would fetch its operands as 2-byte accesses per lane (measured: + 55 % on the launch);
        # linear epilogues + the vectorised pointwise kernels stay the faster form there
        if conv_split.supported([h, *xs], [conv_z, conv_r]):
            z_pre, r_pre = conv_split.conv_nhwc([h, *xs], [conv_z, conv_r])
        else:
            (z_pre,), (r_pre,) = conv_split.conv_nhwc([h, *xs], [conv_z]), conv_split.conv_nhwc([h, *xs], [conv_r])
        (q_pre,) = conv_split.conv_nhwc([gru_reset(h, r_pre, cr), *xs], [conv_q])
        return gru_blend(h, z_pre, q_pre, cz, cq)
    if conv_split.supported([h, *xs], [conv_z, conv_r]):
        z, gated = conv_split.conv_nhwc([h, *xs], [conv_z, conv_r], epilogue=[("sigmoid", cz), ("sigmoid_mul", cr, h)])
    else:   # hidden sizes that do not fill whole 128-channel output tiles: one launch per gate
        (z,) = conv_split.conv_nhwc([h, *xs], [conv_z], epilogue=[("sigmoid", cz)])
        (gated,) = conv_split.conv_nhwc([h, *xs], [conv_r], epilogue=[("sigmoid_mul", cr, h)])
    return conv_split.conv_nhwc([gated, *xs], [conv_q], epilogue=[("gru_blend", cq, h, z)])[0]


class ConvGRU(nn.Module):
    """update.py:16-33.  ``forward(h, cz, cr, cq, *x_list)``: the context tensors are added to the gate
    pre-activations."""

    def __init__(self, hidden_dim, input_dim, kernel_size=3):
        super().__init__()
        for gate in ("z", "r", "q"):
            setattr(self, "conv" + gate, _conv(hidden_dim + input_dim, hidden_dim, kernel_size))

    def forward(self, h, cz, cr, cq, *x_list):
        if _fp16_inference():
            h, cz, cr, cq = _to16(h), _to16(cz), _to16(cr), _to16(cq)
            x_list = [_to16(t) for t in x_list]   # cast the small pieces, not their concatenation
        if (_split_ok(h, cz, cr, cq, *x_list) and conv_split.supported([h, *x_list], [self.convz])
                and conv_split.supported([h, *x_list], [self.convr]) and conv_split.supported([h, *x_list], [self.convq])):
            return _gru_step_split(h, list(x_list), self.convz, self.convr, self.convq, cz, cr, cq)
        return _gru_step(h, torch.cat(x_list, 1), self.convz, self.convr, self.convq, cz, cr, cq)


class SepConvGRU(nn.Module):
    """update.py:35-63: a horizontal (1x5) GRU step followed by a vertical (5x1) one."""

    def __init__(self, hidden_dim=128, input_dim=192 + 128):
        super().__init__()
        for idx, (k, pad) in enumerate((((1, 5), (0, 2)), ((5, 1), (2, 0))), start=1):
            for gate in ("z", "r", "q"):
                setattr(self, f"conv{gate}{idx}", _conv(hidden_dim + input_dim, hidden_dim, k, pad))

    def forward(self, h, *x):
        x = torch.cat(x, 1)
        h = _gru_step(h, x, self.convz1, self.convr1, self.convq1)
        return _gru_step(h, x, self.convz2, self.convr2, self.convq2)


class FlowHead(nn.Module):
    """update.py:6-14: conv3x3 -> ReLU -> conv3x3."""

    def __init__(self, input_dim=128, hidden_dim=256, output_dim=2):
        super().__init__()
        self.conv1 = _conv(input_dim, hidden_dim, 3)
        self.conv2 = _conv(hidden_dim, output_dim, 3)
        self.relu = nn.ReLU(inplace=True)

    def forward(self, x):
        return self.conv2(self.relu(self.conv1(x)))


class BasicMotionEncoder_LZH(nn.Module):
    """update.py:66-91: correlation and flow branches -> 126 channels, flow appended (128 total)."""

    def __init__(self, corr_levels, corr_radius):
        super().__init__()
        self.corr_levels, self.corr_radius = corr_levels, corr_radius
        self.convc1 = _conv(corr_levels * (2 * corr_radius + 1), 64, 1)
        self.convc2 = _conv(64, 64, 3)
        self.convf1 = _conv(2, 64, 7)
        self.convf2 = _conv(64, 64, 3)
        self.conv = _conv(128, 126, 3)

    def forward(self, flow, corr):
        """``corr``: the lookup's ``[B, levels * (2 r + 1), H, W]`` features -- or, on the inference HIP path, the same features
        channels-last and zero-padded to a multiple of 16 channels (``CorrBlock1D.__call__(..., nhwc_channels=)``)."""
        cin = self.convc1.in_channels
        c1 = f1 = None
        if corr.shape[1] >= cin and corr.shape[1] % 16 == 0 and _split_ok(corr) and flow.dtype == torch.float32:
            # the two input convolutions on the HIP path as well: 1x1 over the padded correlation features (matrix cores), 7x7
            # over the 2-channel flow (vector units), ReLU inside
            flow = flow.contiguous()
            if conv_split.supported([corr], [self.convc1], pad_in=True) and conv_split.small_cin_supported(flow, self.convf1):
                (c1,) = conv_split.conv_nhwc([corr], [self.convc1], relu=True, pad_in=True)
                f1 = conv_split.conv_small_cin(flow, self.convf1, relu=True)
        if c1 is None:
            if corr.shape[1] > cin:
                corr = corr[:, :cin]
            c1, f1 = F.relu(self.convc1(corr)), F.relu(self.convf1(flow))   # 1x1 over 36 and 7x7 over 2 channels: the library's
        if (_split_ok(c1, f1) and conv_split.supported([c1], [self.convc2]) and conv_split.supported([f1], [self.convf2])
                and conv_split.supported([c1, f1], [self.conv])):
            (c,) = conv_split.conv_nhwc([c1], [self.convc2], relu=True)
            (f,) = conv_split.conv_nhwc([f1], [self.convf2], relu=True)
            (out,) = conv_split.conv_nhwc([c, f], [self.conv], relu=True, out_channels_total=self.conv.out_channels + flow.shape[1])
            out[:, self.conv.out_channels:] = flow     # the appended flow channels (:91), written in place
            return out
        c = F.relu(self.convc2(c1))
        f = F.relu(self.convf2(f1))
        return torch.cat((F.relu(self.conv(torch.cat((c, f), 1))), flow), 1)


class BasicMultiUpdateBlock_LZH(nn.Module):
    """update.py:130-178.  Three GRU scales (1/4, 1/8, 1/16 of the input for ``n_downsample=2``); returns
    ``(net, mask, delta_flow)`` and, like the reference, writes the new hidden states into ``net``."""

    def __init__(self, corr_levels, corr_radius, n_gru_layers, n_downsample, hidden_dims=[]):
        super().__init__()
        self.n_gru_layers, self.n_downsample = n_gru_layers, n_downsample
        self.encoder = BasicMotionEncoder_LZH(corr_levels=corr_levels, corr_radius=corr_radius)
        h32, h16, h08 = hidden_dims[0], hidden_dims[1], hidden_dims[2]
        self.gru08 = ConvGRU(h08, 128 + h16 * (n_gru_layers > 1))
        self.gru16 = ConvGRU(h16, h32 * (n_gru_layers == 3) + h08)
        self.gru32 = ConvGRU(h32, h16)
        self.flow_head = FlowHead(h08, hidden_dim=256, output_dim=2)
        self.mask = nn.Sequential(_conv(h08, 256, 3), nn.ReLU(inplace=True),
                                  _conv(256, 9 * (2 ** n_downsample) ** 2, 1))

    @staticmethod
    def _like(t, ref):
        """``t`` in the dense layout of ``ref`` (the channels-last loop keeps every concatenation in one layout)."""
        if ref.dim() == 4 and not ref.is_contiguous() and ref.is_contiguous(memory_format=torch.channels_last):
            return t.contiguous(memory_format=torch.channels_last)
        return t

    def forward(self, net, inp, corr=None, flow=None, iter08=True, iter16=True, iter32=True, update=True,
                need_mask=True):
        """``need_mask=False`` (not in the reference) skips the upsampling-mask head and returns ``None`` for it: in
        ``test_mode`` the reference computes the mask in every iteration but only reads the last one
        (``raft_stereo.py:184-185``) -- 13 % of the block's FLOPs."""
        # The motion encoder depends on (flow, corr) only, the coarse GRUs on the hidden states only -- and the coarse GRUs'
        # launches are small (1/16 scale: 288 workgroups for 1024 slots): for inference the encoder runs on a side stream
        # under them (fork / join by events: capturable in a HIP graph).
        motion, side = None, None
        if iter08 and (iter16 or iter32) and _split_ok(net[0]) and flow is not None and flow.is_cuda:
            cur = torch.cuda.current_stream(flow.device)
            side = _side_stream(flow.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                motion = self.encoder(flow, self._like(corr, net[0]))
        if iter32:
            net[2] = self.gru32(net[2], *inp[2], pool2x(net[1]))
        if iter16:
            extra = (self._like(interp(_f32(net[2]), net[1]), net[1]),) if self.n_gru_layers > 2 else ()
            net[1] = self.gru16(net[1], *inp[1], pool2x(net[0]), *extra)
        if iter08:
            extra = (self._like(interp(_f32(net[1]), net[0]), net[0]),) if self.n_gru_layers > 1 else ()
            if side is not None:
                cur.wait_stream(side)
                if not torch.cuda.is_current_stream_capturing():
                    # allocated on the side stream, consumed (and freed) on this one.  (Inside a capture the fork / join above
                    # already orders the block's reuse on the side stream after its last consumer, and record_stream would
                    # defer every iteration's free to the end of the capture: one motion buffer per iteration in the pool.)
                    motion.record_stream(cur)
            else:
                motion = self.encoder(flow, self._like(corr, net[0]))
            net[0] = self.gru08(net[0], *inp[0], motion, *extra)
        if not update:
            return net
        h0 = net[0]
        heads = [self.flow_head.conv1] + ([self.mask[0]] if need_mask else [])
        if _split_ok(h0) and all(conv_split.supported([h0], [m]) for m in heads):
            # flow_head.conv1 | mask[0] read the same hidden state: one launch, ReLU in the epilogue
            if conv_split.supported([h0], heads):
                outs = conv_split.conv_nhwc([h0], heads, relu=True)
            else:
                outs = [conv_split.conv_nhwc([h0], [m], relu=True)[0] for m in heads]
            mask = None
            if need_mask:      # scaled to balance gradients (:176): the factor is folded into the packed 1x1 weights
                if conv_split.supported([outs[1]], [self.mask[2]]):
                    (mask,) = conv_split.conv_nhwc([outs[1]], [self.mask[2]], scales=[0.25])
                else:
                    mask = 0.25 * self.mask[2](outs[1])
            if conv_split.small_supported(outs[0], self.flow_head.conv2):
                return net, mask, conv_split.conv3x3_small(outs[0], self.flow_head.conv2)   # 256 -> 2: fp32 FMAs, NCHW out
            return net, mask, self.flow_head.conv2(outs[0])
        mask = 0.25 * self.mask(net[0]) if need_mask else None  # scaled to balance gradients (:176)
        return net, mask, self.flow_head(net[0])
