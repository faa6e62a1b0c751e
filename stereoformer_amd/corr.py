"""RAFT-style epipolar correlation (A8/A9/A10) behind the reference's surfaces:
``models/IterativeLocalCostVolume/corr/corr.py`` and
``models/IterativeLocalCostVolume/utils/utils.py``."""
from __future__ import annotations

import ctypes

import torch
import torch.nn.functional as F

from . import _native as N


def _level_array(tensors):
    arr = (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
    return arr


def _pyramid_forward(fmap1, fmap2, num_levels):
    B, C, H, W1 = fmap1.shape
    W2 = fmap2.shape[3]
    rows = B * H * W1
    levels = [torch.empty((rows, 1, 1, W2 >> i), device=fmap1.device, dtype=torch.float32) for i in range(num_levels + 1)]
    with torch.cuda.device(fmap1.device):
        N.feat_call("stereo_allpairs_pyramid", [fmap1, fmap2],
                    lambda f: (N.ptr(f[0]), N.ptr(f[1]), _level_array(levels), B, C, H, W1, W2, num_levels,
                               N.stream_ptr(fmap1.device)))
    return levels


class _AllPairsPyramidFn(torch.autograd.Function):
    """All-pairs correlation + pyramid with a HIP backward (pool backward + two f32-MFMA GEMMs)."""

    @staticmethod
    def forward(ctx, fmap1, fmap2, num_levels):
        levels = _pyramid_forward(fmap1, fmap2, num_levels)
        ctx.save_for_backward(fmap1, fmap2)
        ctx.num_levels = num_levels
        return tuple(levels)

    @staticmethod
    def backward(ctx, *glevels):
        f1, f2 = ctx.saved_tensors
        B, C, H, W1 = f1.shape
        W2 = f2.shape[3]
        rows = B * H * W1
        # the kernel folds the coarse cotangents into level 0 in place: work on private copies
        top = max([i for i, g in enumerate(glevels) if g is not None], default=-1)
        if top < 0:
            return None, None, None
        bufs = []
        for i in range(top + 1):
            g = glevels[i]
            bufs.append(N.dev_f32(g, "all-pairs backward: grad").clone() if g is not None else
                        torch.zeros((rows, 1, 1, W2 >> i), device=f1.device, dtype=torch.float32))
        arr = (ctypes.c_void_p * (ctx.num_levels + 1))(*([b.data_ptr() for b in bufs] + [0] * (ctx.num_levels - top)))
        g1 = torch.empty_like(f1) if ctx.needs_input_grad[0] else None
        g2 = torch.empty_like(f2) if ctx.needs_input_grad[1] else None
        with torch.cuda.device(f1.device):
            N.call("stereo_allpairs_pyramid_bwd_f32", N.ptr(f1), N.ptr(f2), arr, N.ptr(g1), N.ptr(g2), B, C, H, W1, W2,
                   ctx.num_levels, N.stream_ptr(f1.device))
        return g1, g2, None


class _CorrLookupFn(torch.autograd.Function):
    """Pyramid lookup with a HIP backward onto the pyramid levels (no gradient w.r.t. coords: the reference
    detaches them every iteration, raft_stereo.py:171)."""

    @staticmethod
    def forward(ctx, coords, shape, radius, num_levels, *levels):
        B, H, W1, W2 = shape
        T = 2 * radius + 1
        out = torch.empty((B, num_levels * T, H, W1), device=coords.device, dtype=torch.float32)
        with torch.cuda.device(coords.device):
            N.call("stereo_corr_lookup_f32", _level_array(list(levels)), N.ptr(coords), N.ptr(out), B, H, W1, W2,
                   num_levels, radius, N.stream_ptr(coords.device))
        ctx.save_for_backward(coords)
        ctx.cfg = (shape, radius, num_levels, [tuple(l.shape) for l in levels])
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (coords,) = ctx.saved_tensors
        (B, H, W1, W2), radius, num_levels, shapes = ctx.cfg
        go = N.dev_f32(grad_out, "corr lookup backward: grad")
        gl = [torch.zeros(sh, device=coords.device, dtype=torch.float32) for sh in shapes]
        with torch.cuda.device(coords.device):
            N.call("stereo_corr_lookup_bwd_f32", _level_array(gl), N.ptr(coords), N.ptr(go), B, H, W1, W2, num_levels,
                   radius, N.stream_ptr(coords.device))
        return (None, None, None, None, *gl)


class CorrBlock1D:
    """``CorrBlock1D(fmap1, fmap2, num_levels=4, radius=4)`` (corr.py:54-111).

    ``corr_pyramid`` holds num_levels+1 tensors [B*H*W1,1,1,W2>>i] like the
    reference (:66-76); ``__call__(coords)`` returns [B, num_levels*(2r+1), H, W1]."""

    def __init__(self, fmap1, fmap2, num_levels=4, radius=4):
        self.num_levels = num_levels
        self.radius = radius
        if fmap2.shape[:3] != fmap1.shape[:3]:
            raise ValueError("CorrBlock1D: fmap1/fmap2 must share B, C, H")
        B, C, H, W1 = fmap1.shape
        self._shape = (B, H, W1, fmap2.shape[3])
        if torch.is_grad_enabled() and (fmap1.requires_grad or fmap2.requires_grad):
            self.corr_pyramid = list(_AllPairsPyramidFn.apply(fmap1.contiguous(), fmap2.contiguous(), num_levels))
            return
        # fp16 feature maps (the fnet's output under autocast, widened by `.float()` in raft_stereo.py:155-156) are taken as
        # they are: the kernel widens them on the way into LDS -- same numbers, half the read bytes
        fmap1 = N.dev_feat(fmap1, "CorrBlock1D: fmap1")
        fmap2 = N.dev_feat(fmap2, "CorrBlock1D: fmap2")
        if fmap1.dtype != fmap2.dtype:
            fmap1, fmap2 = fmap1.float(), fmap2.float()
        N.same_device(fmap1, fmap2)
        self.corr_pyramid = _pyramid_forward(fmap1, fmap2, num_levels)

    def __call__(self, coords, nhwc_channels=None):
        """``nhwc_channels`` (inference; not in the reference): return the ``num_levels * (2 r + 1)`` correlation features as a
        ``torch.channels_last`` tensor zero-padded to that many channels (a multiple of 16) -- the layout the motion encoder's
        first convolution takes on the HIP path -- instead of NCHW."""
        B, H, W1, W2 = self._shape
        coords = N.dev_f32(coords.detach(), "CorrBlock1D: coords")  # no gradient flows to coords (raft_stereo.py:171)
        if coords.dim() != 4 or coords.shape[0] != B or coords.shape[1] < 1 or tuple(coords.shape[2:]) != (H, W1):
            raise ValueError(f"CorrBlock1D: coords must be [B,2,H,W1] = [{B},2,{H},{W1}]")
        if coords.shape[1] != 2:  # the reference only reads channel 0 (corr.py:80)
            coords = torch.cat([coords[:, :1], torch.zeros_like(coords[:, :1])], 1).contiguous()
        if torch.is_grad_enabled() and any(l.requires_grad for l in self.corr_pyramid[:self.num_levels]):
            return _CorrLookupFn.apply(coords, self._shape, self.radius, self.num_levels,
                                       *self.corr_pyramid[:self.num_levels])
        T = 2 * self.radius + 1
        if nhwc_channels is not None:
            if nhwc_channels < self.num_levels * T:
                raise ValueError("CorrBlock1D: nhwc_channels must cover num_levels * (2 * radius + 1)")
            out = torch.empty((B, int(nhwc_channels), H, W1), device=coords.device, dtype=torch.float32,
                              memory_format=torch.channels_last)
            with torch.cuda.device(coords.device):
                N.call("stereo_corr_lookup_nhwc_f32", _level_array(self.corr_pyramid[:self.num_levels]), N.ptr(coords),
                       N.ptr(out), B, H, W1, W2, self.num_levels, self.radius, int(nhwc_channels), N.stream_ptr(coords.device))
            return out
        out = torch.empty((B, self.num_levels * T, H, W1), device=coords.device, dtype=torch.float32)
        with torch.cuda.device(coords.device):
            N.call("stereo_corr_lookup_f32", _level_array(self.corr_pyramid[:self.num_levels]), N.ptr(coords),
                   N.ptr(out), B, H, W1, W2, self.num_levels, self.radius, N.stream_ptr(coords.device))
        return out

    @staticmethod
    def corr(fmap1, fmap2):
        """All-pairs ``einsum('aijk,aijh->ajkh')/sqrt(C)`` -> [B,H,W1,1,W2] (corr.py:103-111)."""
        fmap1 = N.dev_feat(fmap1, "CorrBlock1D.corr: fmap1")
        fmap2 = N.dev_feat(fmap2, "CorrBlock1D.corr: fmap2")
        if fmap1.dtype != fmap2.dtype:
            fmap1, fmap2 = fmap1.float(), fmap2.float()
        B, C, H, W1 = fmap1.shape
        W2 = fmap2.shape[3]
        out = torch.empty((B, H, W1, 1, W2), device=fmap1.device, dtype=torch.float32)
        with torch.cuda.device(fmap1.device):
            N.feat_call("stereo_allpairs_pyramid", [fmap1, fmap2],
                        lambda f: (N.ptr(f[0]), N.ptr(f[1]), _level_array([out]), B, C, H, W1, W2, 0,
                                   N.stream_ptr(fmap1.device)))
        return out


class PytorchAlternateCorrBlock1D:
    """``PytorchAlternateCorrBlock1D`` (corr.py:8-51): the memory-saving variant that never
    builds the all-pairs volume."""

    def __init__(self, fmap1, fmap2, num_levels=4, radius=4):
        self.num_levels = num_levels
        self.radius = radius
        self.corr_pyramid = []
        self._grad_block = None
        if torch.is_grad_enabled() and (fmap1.requires_grad or fmap2.requires_grad):
            # Training: sampling pooled fmap2 and then contracting the channels equals contracting first and sampling the
            # pooled volume (both are linear; SURVEY 8(c) identity I7, 3.7e-5), and that order has HIP backward kernels
            # (A8 + A9).  The memory saving of the alternate block is an inference property.
            if fmap1.shape != fmap2.shape:
                raise ValueError("PytorchAlternateCorrBlock1D: fmap shapes differ")
            self._grad_block = CorrBlock1D(fmap1, fmap2, num_levels=num_levels, radius=radius)
            self.fmap1, self.fmap2 = fmap1, fmap2
            return
        self.fmap1 = N.dev_f32(fmap1, "PytorchAlternateCorrBlock1D: fmap1")
        self.fmap2 = N.dev_f32(fmap2, "PytorchAlternateCorrBlock1D: fmap2")
        if self.fmap1.shape != self.fmap2.shape:
            raise ValueError("PytorchAlternateCorrBlock1D: fmap shapes differ")

    def __call__(self, coords):
        if self._grad_block is not None:
            B, C, H, W = self.fmap1.shape
            rows = torch.arange(H, device=coords.device, dtype=coords.dtype).view(1, H, 1)
            if tuple(coords.shape) != (B, 2, H, W) or not torch.equal(coords[:, 1].detach(), rows.expand(B, H, W)):
                raise RuntimeError("PytorchAlternateCorrBlock1D under grad: coords[:, 1] must be the pixel's own row (what the "
                                   "RAFT loop feeds: delta_flow[:, 1] = 0, raft_stereo.py:180)")
            return self._grad_block(coords)
        coords = N.dev_f32(coords, "PytorchAlternateCorrBlock1D: coords")
        B, C, H, W = self.fmap1.shape
        if tuple(coords.shape) != (B, 2, H, W):
            raise ValueError("PytorchAlternateCorrBlock1D: coords must be [B,2,H,W]")
        T = 2 * self.radius + 1
        out = torch.empty((B, self.num_levels * T, H, W), device=coords.device, dtype=torch.float32)
        nws = int(N.load().stereo_alt_corr_workspace_floats(B, C, H, W, self.num_levels))
        ws = torch.empty((max(nws, 1),), device=coords.device, dtype=torch.float32)
        with torch.cuda.device(coords.device):
            N.call("stereo_alt_corr_lookup_f32", N.ptr(self.fmap1), N.ptr(self.fmap2), N.ptr(coords), N.ptr(out),
                   N.ptr(ws), B, C, H, W, self.num_levels, self.radius, N.stream_ptr(coords.device))
        return out


class AlternateCorrBlock:
    """The reference's CUDA-extension variant is a stub that raises (corr.py:114-116)."""

    def __init__(self, fmap1, fmap2, num_levels=4, radius=4):
        raise NotImplementedError


def bilinear_sampler(img, coords, mode="bilinear", mask=False):
    """``bilinear_sampler`` (utils/utils.py:59-76): pixel coordinates, align_corners=True, zeros
    padding.  img [N,C,H,W], coords [N,Ho,Wo,2]."""
    if mode != "bilinear":
        raise NotImplementedError("bilinear_sampler: only mode='bilinear' (the reference passes it to grid_sample)")
    img = N.dev_f32(img, "bilinear_sampler: img")
    coords = N.dev_f32(coords, "bilinear_sampler: coords")
    n, c, h, w = img.shape
    if coords.dim() != 4 or coords.shape[0] != n or coords.shape[-1] != 2:
        raise ValueError("bilinear_sampler: coords must be [N,Ho,Wo,2]")
    ho, wo = coords.shape[1:3]
    out = torch.empty((n, c, ho, wo), device=img.device, dtype=torch.float32)
    m = torch.empty((n, ho, wo, 1), device=img.device, dtype=torch.float32) if mask else None
    with torch.cuda.device(img.device):
        N.call("stereo_bilinear_sampler_f32", N.ptr(img), N.ptr(coords), N.ptr(out), N.ptr(m), n, c, h, w, ho, wo,
               N.stream_ptr(img.device))
    return (out, m) if mask else out


def coords_grid(batch, ht, wd, device=None):
    """``coords_grid`` (utils/utils.py:79-82): [B,2,H,W], channel 0 = x, channel 1 = y."""
    ys, xs = torch.meshgrid(torch.arange(ht, device=device), torch.arange(wd, device=device), indexing="ij")
    coords = torch.stack([xs, ys], dim=0).float()
    return coords[None].repeat(batch, 1, 1, 1)


class InputPadder:
    """``InputPadder`` (utils/utils.py:7-26): replicate-pads to a multiple of ``divis_by``."""

    def __init__(self, dims, mode="sintel", divis_by=8):
        self.ht, self.wd = dims[-2:]
        pad_ht = (((self.ht // divis_by) + 1) * divis_by - self.ht) % divis_by
        pad_wd = (((self.wd // divis_by) + 1) * divis_by - self.wd) % divis_by
        if mode == "sintel":
            self._pad = [pad_wd // 2, pad_wd - pad_wd // 2, pad_ht // 2, pad_ht - pad_ht // 2]
        else:
            self._pad = [pad_wd // 2, pad_wd - pad_wd // 2, 0, pad_ht]

    def pad(self, *inputs):
        assert all((x.ndim == 4) for x in inputs)
        return [F.pad(x, self._pad, mode="replicate") for x in inputs]

    def unpad(self, x):
        assert x.ndim == 4
        ht, wd = x.shape[-2:]
        c = [self._pad[2], ht - self._pad[3], self._pad[0], wd - self._pad[1]]
        return x[..., c[0]:c[1], c[2]:c[3]]
