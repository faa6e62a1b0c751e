"""Stride-1 "same" convolutions of the update block (A14, ``update/update.py:6-33,66-91,130-178``) on the HIP path.

``conv_nhwc(srcs, convs, relu)`` is ``[m(torch.cat(srcs, 1)) for m in convs]`` (each ``m`` an ``nn.Conv2d`` with a 3x3 or 1x1
kernel, stride 1, padding k // 2, the same input) for channels-last fp32 activations, as ONE launch of
``csrc/conv_split.hip``: the concatenation is never materialised (the kernel's K loop walks the sources), the bias and an
optional ReLU ride in the epilogue, and several modules that read the same input (``convz`` | ``convr``; ``flow_head.conv1`` |
``mask[0]``) run as one convolution whose output channels are scattered to one tensor per module.  Arithmetic: fp32-faithful
split-bf16 on ``v_mfma_f32_32x32x16_bf16`` (see the kernel's header).

Inference only (no autograd formula: the training path keeps the library convolutions); ``supported(...)`` tells a caller whether
its tensors are inside the kernel's envelope."""
from __future__ import annotations

import ctypes

import torch

from . import _native as N

_PRECISION = "highest"


def set_precision(precision):
    """Arithmetic of ``conv_nhwc`` on fp32 activations: ``"highest"`` (default, the parity configuration: three bf16 pieces per
    value, six piece products -- fp32-faithful) or ``"high"`` (two pieces, three products, "bf16x3": ~2^-16 relative error per
    product -- the counterpart of ``torch.set_float32_matmul_precision('high')`` -- at half the matrix work).  Returns the
    previous setting."""
    global _PRECISION
    if precision not in ("highest", "high"):
        raise ValueError("conv_split.set_precision: 'highest' or 'high'")
    prev, _PRECISION = _PRECISION, precision
    return prev


def _is_cl(t):
    return t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last)


def supported(srcs, convs) -> bool:
    """Channels-last fp32 (or all-fp16: the mixed_precision form) HIP tensors without grad, every source channel count a multiple of 16 (<= 4 sources), 3x3 / 1x1 stride-1
    'same' convolutions without groups or dilation, all but the last module with a multiple of 128 output channels, <= 512 in
    all."""
    if not (1 <= len(srcs) <= 4) or not convs:
        return False
    if torch.is_grad_enabled() and (any(t.requires_grad for t in srcs) or
                                    any(p.requires_grad for m in convs if isinstance(m, torch.nn.Module) for p in m.parameters())):
        return False     # no autograd formula: training keeps the library convolutions
    t0 = srcs[0]
    if not isinstance(t0, torch.Tensor) or t0.dtype not in (torch.float32, torch.float16):
        return False
    for t in srcs:
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == t0.dtype and _is_cl(t)):
            return False
        if t.shape[0] != t0.shape[0] or t.shape[2:] != t0.shape[2:] or t.shape[1] % 16 != 0 or t.shape[1] < 16:
            return False
        if t.shape[1] * t.shape[2] * t.shape[3] * 4 >= 2 ** 31:
            return False
    cin = sum(t.shape[1] for t in srcs)
    k = convs[0].kernel_size
    total = 0
    for i, m in enumerate(convs):
        if not isinstance(m, torch.nn.Conv2d) or m.kernel_size != k or k not in ((3, 3), (1, 1)):
            return False
        if m.stride != (1, 1) or m.dilation != (1, 1) or m.groups != 1 or m.padding != (k[0] // 2, k[0] // 2):
            return False
        if m.in_channels != cin or m.weight.dtype != torch.float32 or m.padding_mode != "zeros":
            return False
        if i + 1 < len(convs) and m.out_channels % 128 != 0:
            return False
        total += m.out_channels
    return total <= 512


def _signature(convs, device):
    """What a packed copy depends on: the identity, storage address and in-place version of every parameter involved."""
    sig = [device.index]
    for m in convs:
        for p in (m.weight, m.bias):
            sig.append(None if p is None else (id(p), p.data_ptr(), p._version, tuple(p.shape)))
    return tuple(sig)


def _cached(convs, tag, device, build):
    """Per-module cache (it lives on the first module of the group and dies with it -- a global table keyed by storage
    addresses would hand a freed weight's packed copy to whatever tensor is allocated at the same address next)."""
    store = convs[0].__dict__.setdefault("_stereo_conv_cache", {})
    key = (tag, device.index) + tuple(id(m) for m in convs[1:])   # (per device: nn.DataParallel replicas share this dict)
    sig = _signature(convs, device)
    hit = store.get(key)
    if hit is not None and hit[0] == sig:
        return hit[1]
    val = build()
    store[key] = (sig, val)
    return val


def _packed(convs, device, sfx=""):
    """Packed weights (split-bf16 pieces for fp32 activations -- three, or two with ``sfx="_bf16x3"`` --, fp16 for fp16 ones,
    ``sfx="_f16"``; + the concatenated fp32 bias) of a group of modules; rebuilt when a parameter is replaced or updated in place."""

    def build():
        with torch.no_grad():
            w = torch.cat([m.weight.detach().contiguous() for m in convs], 0).contiguous()
            bias = None
            if any(m.bias is not None for m in convs):
                bias = torch.cat([m.bias.detach() if m.bias is not None else torch.zeros(m.out_channels, device=device)
                                  for m in convs]).contiguous()
            cout, cin, k = w.shape[0], w.shape[1], w.shape[2]
            nbytes = getattr(N.load(), "stereo_conv_packed_bytes" + sfx)(cout, cin, k)
            if nbytes <= 0:
                raise ValueError(f"conv_nhwc: unsupported weight shape {tuple(w.shape)}")
            packed = torch.empty(nbytes, device=device, dtype=torch.uint8)
            with torch.cuda.device(device):
                N.call("stereo_conv_pack_weights" + sfx, N.ptr(w), N.ptr(packed), cout, cin, k, N.stream_ptr(device))
        return packed, bias, k
    return _cached(convs, "split" + sfx, device, build)


def conv_nhwc(srcs, convs, relu=False, out_channels_total=None):
    """``[m(torch.cat(srcs, 1)) for m in convs]`` (optionally followed by ReLU), channels-last fp32 (or fp16: what the modules give
    under fp16 autocast) in and out, one launch.
    ``out_channels_total`` (single module only): allocate the output with that many channels and fill the first
    ``out_channels`` of them -- the caller owns the rest (the motion encoder appends the 2 flow channels, update.py:89-91)."""
    srcs = list(srcs)
    convs = list(convs)
    if not supported(srcs, convs):
        raise ValueError("conv_nhwc: operands outside the kernel's envelope (see conv_split.supported)")
    if out_channels_total is not None and (len(convs) != 1 or out_channels_total < convs[0].out_channels):
        raise ValueError("conv_nhwc: out_channels_total applies to one module and must cover its output channels")
    dev = srcs[0].device
    half = srcs[0].dtype == torch.float16
    sfx = "_f16" if half else ("_bf16x3" if _PRECISION == "high" else "")
    packed, bias, k = _packed(convs, dev, sfx)
    B, _, H, W = srcs[0].shape
    widths = [m.out_channels for m in convs] if out_channels_total is None else [int(out_channels_total)]
    outs = [torch.empty((B, c, H, W), device=dev, dtype=srcs[0].dtype, memory_format=torch.channels_last) for c in widths]
    sp = (N._vp * len(srcs))(*[t.data_ptr() for t in srcs])
    sc = (N._i * len(srcs))(*[t.shape[1] for t in srcs])
    op = (N._vp * len(outs))(*[t.data_ptr() for t in outs])
    oc = (N._i * len(outs))(*[m.out_channels for m in convs])
    os_ = (N._i * len(outs))(*widths)
    with torch.cuda.device(dev):
        N.call("stereo_conv_nhwc_f16" if half else "stereo_conv_nhwc_f32" + sfx, sp, sc, len(srcs), N.ptr(packed), N.ptr(bias), op, oc, os_,
               len(outs), k,
               1 if relu else 0, B, H, W, N.stream_ptr(dev))
    return outs


def small_supported(x, conv) -> bool:
    """3x3 stride-1 'same' convolution with <= 4 output channels over a channels-last fp32 HIP tensor, Cin % 32 == 0."""
    if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in conv.parameters())):
        return False
    return (isinstance(conv, torch.nn.Conv2d) and conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1)
            and conv.dilation == (1, 1) and conv.groups == 1 and conv.padding_mode == "zeros" and conv.out_channels <= 4
            and conv.weight.dtype == torch.float32 and x.is_cuda and x.dtype in (torch.float32, torch.float16) and _is_cl(x)
            and x.shape[1] == conv.in_channels and x.shape[1] % 32 == 0 and x.shape[1] * x.shape[2] * x.shape[3] * 4 < 2 ** 31)


def conv3x3_small(x, conv):
    """``conv(x)`` for a 3x3 convolution with <= 4 output channels (``flow_head.conv2``): channels-last fp32 / fp16 in,
    NCHW-contiguous fp32 out."""
    if not small_supported(x, conv):
        raise ValueError("conv3x3_small: operands outside the kernel's envelope (see conv_split.small_supported)")
    half = x.dtype == torch.float16   # mixed_precision: the weights rounded to fp16 as autocast hands them to the library
    wt = _cached([conv], "tco16" if half else "tco", x.device,
                 lambda: (conv.weight.detach().half().float() if half else conv.weight.detach()).permute(2, 3, 1, 0).contiguous())
    B, C, H, W = x.shape
    out = torch.empty((B, conv.out_channels, H, W), device=x.device, dtype=torch.float32)
    with torch.cuda.device(x.device):
        N.call("stereo_conv3x3_small_nhwc_f16" if half else "stereo_conv3x3_small_nhwc_f32", N.ptr(x), N.ptr(wt),
               N.ptr(None if conv.bias is None else conv.bias.detach()),
               N.ptr(out), B, C, conv.out_channels, H, W, N.stream_ptr(x.device))
    return out
