"""Re-hosted hot loops of the reference models (the callers either side of the kernels):

* ``RAFTStereoLoop``  -- ``RAFT_Stereo.forward`` from the correlation block on
  (``models/IterativeLocalCostVolume/raft_stereo.py:150-199``): all-pairs pyramid (A8), then per
  iteration lookup (A9) -> ``BasicMultiUpdateBlock_LZH`` (A14) -> coordinate update, convex
  upsampling (A13) on the iterations the reference upsamples.
* ``LowCNNGRULoop``   -- the refinement loop of ``LowCNN`` (``models/LocalCostVolume/baseline_dynamic_gru.py:128-137``):
  16 x (``DisparityUpdateDLCWithMask`` (A11) -> ``upsample_convex8`` (A13)).

The feature / context extractors that feed them are out of scope (plain convolutions, SURVEY 2 rows
13/16/18).  Both loops run without any host synchronisation (the reference's in-loop asserts and
debug print are gone) and can replay one captured iteration as a HIP graph (``use_graph=True``),
which removes the ~60 launches per iteration from the host's critical path.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _native
from .corr import CorrBlock1D, coords_grid
from .update_dlc import DisparityUpdateDLCWithMask
from .update_raft import BasicMultiUpdateBlock_LZH
from .upsample import upflow8, upsample_convex8, upsample_flow


class _GraphCache:
    """Whole-forward HIP-graph capture keyed by input shapes: inputs are copied into static buffers,
    the captured launch sequence (every kernel goes to the capturing stream: the C ABI takes the
    stream explicitly and never synchronises) is replayed, static outputs are returned as clones."""

    def __init__(self):
        self._entries = {}

    def clear(self):
        """Forget every captured graph (their launches read packed weight copies that have been rebuilt)."""
        self._entries.clear()

    def run(self, key, fn, tensors):
        ent = self._entries.get(key)
        if ent is None:
            static_in = [t.clone() for t in tensors]
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):               # warm-up outside capture: MIOpen picks algorithms here
                for _ in range(2):
                    fn(static_in)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                static_out = fn(static_in)
            ent = (graph, static_in, static_out)
            self._entries[key] = ent
        graph, static_in, static_out = ent
        for s, t in zip(static_in, tensors):
            s.copy_(t)
        graph.replay()
        return [o.clone() for o in static_out]


class RAFTStereoLoop(nn.Module):
    def __init__(self, hidden_dims=(128, 128, 128), downsample=2, corr_levels=4, corr_radius=4, n_gru_layers=3,
                 mixed_precision=False, channels_last=None, conv_precision="highest"):
        """``channels_last``: keep the update block's weights and activations in NHWC, where the block's convolutions run on the
        library's own kernels (``conv_split``: gate math in the epilogues, no concatenations) for inference.  ``None`` (default):
        NHWC whenever the forward runs without gradients -- the drop-in inference path IS the HIP path --, the reference's
        NCHW + library convolutions under autograd; ``True`` / ``False`` force one layout for both.
        ``mixed_precision``: the reference's flag of the same name (``raft_stereo.py:176``): the update block's
        convolutions run under fp16 ``autocast``; correlation, lookup, gate math, coordinates and the convex
        upsampling stay fp32.  Off by default -- fp32 is the parity configuration.
        ``conv_precision`` (inference in NHWC): ``"highest"`` = fp32-faithful split-fp16 arithmetic (default, parity;
        ``conv_split``'s docstring), ``"bf16x6"`` = the split-bf16 form with fp32's whole exponent range (twice the matrix work),
        ``"high"`` = two bf16 pieces / three products per fp32 product (like ``torch.set_float32_matmul_precision('high')``):
        ~2^-16 relative error per product."""
        super().__init__()
        if conv_precision not in ("highest", "f16x3", "f16x3s", "bf16x6", "high", "bf16x3"):
            raise ValueError("RAFTStereoLoop: conv_precision is 'highest' (fp32-faithful, the parity configuration), 'bf16x6' or 'high'")
        self.conv_precision = conv_precision
        self.mixed_precision = bool(mixed_precision)
        self.channels_last = None if channels_last is None else bool(channels_last)
        self.hidden_dims = list(hidden_dims)
        self.downsample = downsample
        self.corr_levels = corr_levels
        self.corr_radius = corr_radius
        self.n_gru_layers = n_gru_layers
        self.update_block = BasicMultiUpdateBlock_LZH(corr_levels=corr_levels, corr_radius=corr_radius,
                                                      n_gru_layers=n_gru_layers, n_downsample=downsample,
                                                      hidden_dims=self.hidden_dims)
        self._graphs = _GraphCache()

    def _iteration(self, corr_fn, coords0, coords1, net_list, inp_list, need_mask=True):
        coords1 = coords1.detach()                                  # raft_stereo.py:171
        if self._nhwc() and not self.mixed_precision and not torch.is_grad_enabled() and isinstance(corr_fn, CorrBlock1D):
            # the HIP path's layout for the motion encoder's 1x1 convolution: channels-last, zero-padded to 16 n channels
            corr = corr_fn(coords1, nhwc_channels=(self.corr_levels * (2 * self.corr_radius + 1) + 15) // 16 * 16)
        else:
            corr = corr_fn(coords1)                                 # :173
        flow = coords1 - coords0
        with torch.autocast("cuda", dtype=torch.float16, enabled=self.mixed_precision):    # :176
            net_list, up_mask, delta_flow = self.update_block(net_list, inp_list, corr, flow,
                                                              iter32=self.n_gru_layers == 3,
                                                              iter16=self.n_gru_layers >= 2, need_mask=need_mask)
        delta_flow = delta_flow.float()
        delta_flow[:, 1] = 0.0                                      # :180
        return net_list, None if up_mask is None else up_mask.float(), coords1 + delta_flow

    def forward(self, fmap1, fmap2, net_list, inp_list, iters=12, flow_init=None, test_mode=False, use_graph=False):
        """fmap1/fmap2 [B,256,H/4,W/4] (fnet outputs), net_list: tanh'ed hidden states per scale,
        inp_list: per scale the (cz, cr, cq) context tensors (``raft_stereo.py:143-148``).
        ``use_graph``: replay the whole loop as one captured HIP graph (same numbers, no per-kernel
        host launch cost)."""
        if not use_graph:
            return self._forward_eager(fmap1, fmap2, net_list, inp_list, iters, flow_init, test_mode)
        if torch.is_grad_enabled():
            raise RuntimeError("RAFTStereoLoop: use_graph=True is an inference path; call it under torch.no_grad()")
        nl, ni = len(net_list), [len(x) for x in inp_list]
        flat = [fmap1, fmap2, *net_list, *[t for lvl in inp_list for t in lvl]]
        if flow_init is not None:
            flat.append(flow_init)

        def fn(ts):
            nets = list(ts[2:2 + nl])
            rest, inps = ts[2 + nl:], []
            for n in ni:
                inps.append(list(rest[:n]))
                rest = rest[n:]
            out = self._forward_eager(ts[0], ts[1], nets, inps, iters, rest[0] if flow_init is not None else None,
                                      test_mode)
            return list(out)

        key = (tuple(tuple(t.shape) for t in flat), iters, bool(test_mode), flat[0].device.index, self.mixed_precision,
               self._nhwc(), self.conv_precision)
        from . import conv_split
        if conv_split.refresh(self.update_block):
            self._graphs.clear()
        out = self._graphs.run(key, fn, flat)
        return tuple(out) if test_mode else out

    def _nhwc(self):
        return (not torch.is_grad_enabled()) if self.channels_last is None else self.channels_last

    def _forward_eager(self, fmap1, fmap2, net_list, inp_list, iters, flow_init, test_mode):
        from . import conv_split
        if not torch.cuda.is_current_stream_capturing() and conv_split.refresh(self.update_block):
            self._graphs.clear()     # weights changed under a packed copy: captured launches point at the old one
        prev = conv_split.set_precision(self.conv_precision)
        try:
            return self._forward_eager_impl(fmap1, fmap2, net_list, inp_list, iters, flow_init, test_mode)
        finally:
            conv_split.set_precision(prev)

    def _forward_eager_impl(self, fmap1, fmap2, net_list, inp_list, iters, flow_init, test_mode):
        # raft_stereo.py:155-156 widens the feature maps with `.float()`; fp16 maps are handed over as they are -- the
        # all-pairs kernel widens them on the way into LDS (same numbers, half the read bytes)
        keep = (torch.float16, torch.float32)
        corr_fn = CorrBlock1D(fmap1 if fmap1.dtype in keep else fmap1.float(), fmap2 if fmap2.dtype in keep else fmap2.float(),
                              radius=self.corr_radius, num_levels=self.corr_levels)
        N, _, H, W = net_list[0].shape
        coords0 = coords_grid(N, H, W, device=fmap1.device)
        coords1 = coords0.clone()
        if flow_init is not None:
            coords1 = coords1 + flow_init
        net_list = [n.contiguous() for n in net_list]   # a private list: the update block writes into it
        if self.mixed_precision and not torch.is_grad_enabled():
            inp_list = [[t.half() for t in lvl] for lvl in inp_list]   # context biases: cast once, not per iteration
        if self._nhwc():
            cl = torch.channels_last
            self.update_block.to(memory_format=cl)   # no-op once converted
            net_list = [_native.nhwc(n) for n in net_list]
            inp_list = [[_native.nhwc(t) for t in lvl] for lvl in inp_list]
        elif self.channels_last is None:
            self.update_block.to(memory_format=torch.contiguous_format)   # back from an earlier inference call (no-op otherwise)
        preds = []
        up_mask = None
        for itr in range(iters):
            skip_up = test_mode and itr < iters - 1   # raft_stereo.py:184-185: only the last mask is read
            net_list, up_mask, coords1 = self._iteration(corr_fn, coords0, coords1, net_list, inp_list,
                                                         need_mask=not skip_up)
            if skip_up:
                continue
            if up_mask is None:
                flow_up = upflow8(coords1 - coords0)
            else:
                flow_up = upsample_flow(coords1 - coords0, up_mask, self.downsample)
            preds.append(flow_up[:, :1])
        if test_mode:
            return coords1 - coords0, preds[-1]
        return preds

class LowCNNGRULoop(nn.Module):
    def __init__(self, input_channels=24, hidden_dim=32, sample_points=20, channels_last=None):
        """``channels_last``: keep the updater's weights and activations in NHWC; for inference every convolution of the updater
        (encoder with BatchNorm folded and zero-padded inputs, GRU gates with the gate math in their epilogues, offset head,
        mask head incl. its 1x1 projection) then runs on this package's own convolution kernels (``conv_split``).
        Same arithmetic to fp32 rounding.  ``None`` (default): NHWC whenever the forward runs without gradients, NCHW + library
        convolutions under autograd."""
        super().__init__()
        self.update = DisparityUpdateDLCWithMask(input_channels=input_channels, hidden_dim=hidden_dim,
                                                 sample_points=sample_points)
        self.channels_last = None if channels_last is None else bool(channels_last)
        self._graphs = _GraphCache()

    def _nhwc(self):
        return (not torch.is_grad_enabled()) if self.channels_last is None else self.channels_last

    def forward(self, cost_volume, disp, left_image, right_image, iters=12, use_graph=False, graph_all_iterates=False):
        """cost_volume [B,D,h,w] (1/8 scale), disp [B,1,h,w] -> list of full-resolution disparities
        (one per iteration, like the reference), or only the last one when ``use_graph`` (all of them with
        ``graph_all_iterates``).  Differentiable in grad mode (every custom op on the path has a HIP backward); the
        graph replay is inference-only."""
        B, D, h, w = cost_volume.shape
        hidden = torch.zeros(B, 2 * self.update.hidden_dim, h, w, device=cost_volume.device)
        from . import conv_split
        if not torch.cuda.is_current_stream_capturing() and conv_split.refresh(self.update):
            self._graphs.clear()
        if self._nhwc():
            self.update.to(memory_format=torch.channels_last)   # no-op once converted
            hidden = hidden.contiguous(memory_format=torch.channels_last)
        elif self.channels_last is None:
            self.update.to(memory_format=torch.contiguous_format)         # back from an earlier inference call
        # the images at the disparity's resolution are loop-invariant: resized once here instead of once per iteration inside
        # BasicGuidanceNet (Attempts/update.py:76-81 -- same kernel, same values)
        from .update_dlc import resize_bilinear

        def small(img):
            return img if img.shape[-1] == w and img.shape[-2] == h else resize_bilinear(img, (h, w), align_corners=False)
        if use_graph:
            if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
                raise RuntimeError("LowCNNGRULoop: use_graph=True is an inference path; call it under torch.no_grad()")

            def fn(ts):
                d, hid = ts[1], torch.zeros_like(hidden)   # (preserves the channels-last layout)
                ts = [ts[0], ts[1], small(ts[2]), small(ts[3])]
                outs = []
                for it in range(iters):   # unless every iterate is wanted, only the last one is upsampled: the
                    last = it == iters - 1  # earlier mask heads are skipped
                    d, hid, mask = self.update(ts[0], d, ts[2], ts[3], hid, True, need_mask=last or graph_all_iterates)
                    if last or graph_all_iterates:
                        outs.append(upsample_convex8(d, mask))
                return outs

            key = (tuple(cost_volume.shape), tuple(left_image.shape), iters, cost_volume.device.index,
                   bool(graph_all_iterates), self._nhwc())
            return self._graphs.run(key, fn, [cost_volume, disp, left_image, right_image])
        outs = []
        left_image, right_image = small(left_image), small(right_image)
        for _ in range(iters):
            disp, hidden, mask = self.update(cost_volume, disp, left_image, right_image, hidden, True)
            outs.append(upsample_convex8(disp, mask))
        return outs
