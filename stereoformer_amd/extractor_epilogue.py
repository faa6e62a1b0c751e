"""Extractor epilogue for the feature hand-over.

The reference's extractors close their stages with ``nn.InstanceNorm2d`` followed by ``ReLU``
(``models/IterativeLocalCostVolume/extractor/extractor.py:135,160-162,187-188``) and hand fp32
NCHW features to the cost-volume ops (``raft_stereo.py:155-156``, ``models/LocalCostVolume/baseline.py:89-114``).
``InstanceNormReLU`` is that pair as ONE op on the HIP path, writing the features in the precision the
matrix-core loaders take (fp16 halves the read side of every cost-volume launch; fp32 keeps bit-level
parity with the reference chain).  The convolutions themselves stay with MIOpen (out of scope).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _native as N


def _channels_last_operand(x):
    """True when ``x`` can go to the channels-last kernel as it is: a HIP [B,C,H,W] tensor in torch.channels_last memory (and
    not also NCHW-contiguous), fp32 or fp16, C a multiple of 4 with 256 % (C / 4) == 0, C <= 256, H*W a multiple of 4."""
    if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dim() == 4 and x.dtype in (torch.float32, torch.float16)):
        return False
    if torch.is_grad_enabled() and x.requires_grad:
        return False
    if x.is_contiguous() or not x.is_contiguous(memory_format=torch.channels_last):
        return False
    C, hw = x.shape[1], x.shape[2] * x.shape[3]
    return C % 4 == 0 and C <= 256 and 256 % (C // 4) == 0 and hw % 4 == 0


def instance_norm_relu(x, weight=None, bias=None, eps=1e-5, relu=True, out_dtype=torch.float16):
    """``relu(F.instance_norm(x, weight=weight, bias=bias, eps=eps))`` -> ``out_dtype`` (fp16 or fp32), NCHW.
    A ``torch.channels_last`` input (fp32 or fp16 -- what MIOpen's convolutions hand over) is read in place: the transpose to
    the NCHW layout of the cost-volume loaders happens inside the normalisation's second pass.
    Forward-only (the extractor is out of scope for training parity): raises under grad like the other forward-only ops."""
    if out_dtype not in (torch.float16, torch.float32):
        raise TypeError("instance_norm_relu: out_dtype must be torch.float16 or torch.float32")
    nhwc = _channels_last_operand(x)
    if not nhwc:
        if isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float16:
            x = x.float()   # fp16 activations outside the channels-last kernel's envelope: widened (one extra pass)
        x = N.dev_f32(x, "instance_norm_relu: x")
    if x.dim() != 4:
        raise ValueError("instance_norm_relu: expected [B,C,H,W]")
    B, C, H, W = x.shape
    w = None if weight is None else N.dev_f32(weight, "instance_norm_relu: weight")
    b = None if bias is None else N.dev_f32(bias, "instance_norm_relu: bias")
    if (w is not None and w.numel() != C) or (b is not None and b.numel() != C):
        raise ValueError("instance_norm_relu: weight / bias must have C elements")
    out = torch.empty((B, C, H, W), device=x.device, dtype=out_dtype)
    lib = N.load()
    ws = torch.empty(max(1, lib.stereo_instance_norm_workspace_bytes(B, C) // 8), device=x.device, dtype=torch.float64)
    with torch.cuda.device(x.device):
        if nhwc:   # channels-last activations (fp32 / fp16) straight from the convolution: no .contiguous() pass
            try:
                N.call("stereo_instance_norm_relu_nhwc", N.ptr(x), 1 if x.dtype == torch.float16 else 0, N.ptr(w), N.ptr(b),
                       N.ptr(out), N.ptr(ws), 1 if out_dtype == torch.float16 else 0, B, C, H, W, float(eps), 1 if relu else 0,
                       N.stream_ptr(x.device))
                return out
            except N.StereoHipError as e:
                if e.code != N.ERR_UNSUPPORTED:
                    raise
                # outside the channels-last kernel's envelope (channel count, H*W % 4, a misaligned view): bring the tensor to
                # NCHW fp32 -- the tiled relayout kernel where it applies, else contiguous() -- and take the planar kernel
                x = N.dev_f32(x.float() if x.dtype == torch.float16 else x, "instance_norm_relu: x")
        try:
            N.call("stereo_instance_norm_relu_f32", N.ptr(x), N.ptr(w), N.ptr(b), N.ptr(out), N.ptr(ws),
                   1 if out_dtype == torch.float16 else 0, B, C, H, W, float(eps), 1 if relu else 0, N.stream_ptr(x.device))
        except N.StereoHipError as e:
            if e.code != N.ERR_UNSUPPORTED:
                raise
            raise N.StereoHipError(f"instance_norm_relu: the planar kernel needs H*W = {H * W} to be a multiple of 4 and 16-byte "
                                   "aligned tensors (16-byte plane rows); pad the image like the reference's InputPadder does",
                                   e.code) from None
    return out


class InstanceNormReLU(nn.Module):
    """``nn.Sequential(nn.InstanceNorm2d(num_features, eps, affine=affine), nn.ReLU())`` as one HIP op.

    Parameter names follow ``nn.InstanceNorm2d`` (``weight`` / ``bias`` when ``affine``), so a reference ``state_dict`` entry
    for the norm layer loads unchanged.  ``out_dtype``: torch.float16 (default; what the ``_f16`` cost-volume loaders take) or
    torch.float32."""

    def __init__(self, num_features, eps=1e-5, affine=False, relu=True, out_dtype=torch.float16):
        super().__init__()
        self.num_features, self.eps, self.affine, self.relu, self.out_dtype = num_features, eps, affine, relu, out_dtype
        if affine:
            self.weight = nn.Parameter(torch.ones(num_features))
            self.bias = nn.Parameter(torch.zeros(num_features))
        else:
            self.register_parameter("weight", None)
            self.register_parameter("bias", None)

    def forward(self, x):
        if x.dim() != 4 or x.shape[1] != self.num_features:
            raise ValueError(f"InstanceNormReLU: expected [B,{self.num_features},H,W], got {tuple(x.shape)}")
        w = None if self.weight is None else self.weight.detach()
        b = None if self.bias is None else self.bias.detach()
        return instance_norm_relu(x, w, b, self.eps, self.relu, self.out_dtype)
