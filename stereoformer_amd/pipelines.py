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

from .corr import CorrBlock1D, coords_grid
from .update_dlc import DisparityUpdateDLCWithMask
from .update_raft import BasicMultiUpdateBlock_LZH
from .upsample import upflow8, upsample_convex8, upsample_flow


class RAFTStereoLoop(nn.Module):
    def __init__(self, hidden_dims=(128, 128, 128), downsample=2, corr_levels=4, corr_radius=4, n_gru_layers=3):
        super().__init__()
        self.hidden_dims = list(hidden_dims)
        self.downsample = downsample
        self.corr_levels = corr_levels
        self.corr_radius = corr_radius
        self.n_gru_layers = n_gru_layers
        self.update_block = BasicMultiUpdateBlock_LZH(corr_levels=corr_levels, corr_radius=corr_radius,
                                                      n_gru_layers=n_gru_layers, n_downsample=downsample,
                                                      hidden_dims=self.hidden_dims)
        self._graph = None

    def _iteration(self, corr_fn, coords0, coords1, net_list, inp_list):
        corr = corr_fn(coords1)                                     # raft_stereo.py:173
        flow = coords1 - coords0
        net_list, up_mask, delta_flow = self.update_block(net_list, inp_list, corr, flow,
                                                          iter32=self.n_gru_layers == 3,
                                                          iter16=self.n_gru_layers >= 2)
        delta_flow[:, 1] = 0.0                                      # :180
        return net_list, up_mask, coords1 + delta_flow

    @torch.no_grad()
    def forward(self, fmap1, fmap2, net_list, inp_list, iters=12, flow_init=None, test_mode=False, use_graph=False):
        """fmap1/fmap2 [B,256,H/4,W/4] (fnet outputs), net_list: tanh'ed hidden states per scale,
        inp_list: per scale the (cz, cr, cq) context tensors (``raft_stereo.py:143-148``)."""
        corr_fn = CorrBlock1D(fmap1.float(), fmap2.float(), radius=self.corr_radius, num_levels=self.corr_levels)
        N, _, H, W = net_list[0].shape
        coords0 = coords_grid(N, H, W, device=fmap1.device)
        coords1 = coords0.clone()
        if flow_init is not None:
            coords1 = coords1 + flow_init
        net_list = [n.contiguous() for n in net_list]
        preds = []
        up_mask = None
        if use_graph and iters > 2:
            coords1, net_list, up_mask = self._run_graphed(corr_fn, coords0, coords1, net_list, inp_list, iters)
            flow_up = upsample_flow(coords1 - coords0, up_mask, self.downsample)[:, :1]
            if test_mode:
                return coords1 - coords0, flow_up
            return [flow_up]
        for itr in range(iters):
            net_list, up_mask, coords1 = self._iteration(corr_fn, coords0, coords1, net_list, inp_list)
            if test_mode and itr < iters - 1:
                continue
            if up_mask is None:
                flow_up = upflow8(coords1 - coords0)
            else:
                flow_up = upsample_flow(coords1 - coords0, up_mask, self.downsample)
            preds.append(flow_up[:, :1])
        if test_mode:
            return coords1 - coords0, preds[-1]
        return preds

    def _run_graphed(self, corr_fn, coords0, coords1, net_list, inp_list, iters):
        """Capture ONE iteration on static buffers and replay it; returns the final state.  (test_mode
        semantics: only the last iteration is upsampled, raft_stereo.py:184-185.)"""
        s_coords = coords1.clone()
        s_net = [n.clone() for n in net_list]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                               # warm-up: MIOpen picks its algorithms here
            for _ in range(2):
                nl, mask, c1 = self._iteration(corr_fn, coords0, s_coords, list(s_net), inp_list)
            s_coords.copy_(coords1)
            for a, b in zip(s_net, net_list):
                a.copy_(b)
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            nl, s_mask, c1 = self._iteration(corr_fn, coords0, s_coords, list(s_net), inp_list)
            s_coords.copy_(c1)
            for a, b in zip(s_net, nl):
                a.copy_(b)
        for _ in range(iters):
            g.replay()
        return s_coords, s_net, s_mask


class LowCNNGRULoop(nn.Module):
    def __init__(self, input_channels=24, hidden_dim=32, sample_points=20):
        super().__init__()
        self.update = DisparityUpdateDLCWithMask(input_channels=input_channels, hidden_dim=hidden_dim,
                                                 sample_points=sample_points)

    @torch.no_grad()
    def forward(self, cost_volume, disp, left_image, right_image, iters=12, use_graph=False):
        """cost_volume [B,D,h,w] (1/8 scale), disp [B,1,h,w] -> list of full-resolution disparities
        (one per iteration, like the reference), or only the last one when ``use_graph``."""
        B, D, h, w = cost_volume.shape
        hidden = torch.zeros(B, 2 * self.update.hidden_dim, h, w, device=cost_volume.device)
        if use_graph and iters > 2:
            s_disp, s_hidden = disp.clone(), hidden.clone()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    self.update(cost_volume, s_disp, left_image, right_image, s_hidden, True)
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                d2, h2, s_mask = self.update(cost_volume, s_disp, left_image, right_image, s_hidden, True)
                s_disp.copy_(d2)
                s_hidden.copy_(h2)
            for _ in range(iters):
                g.replay()
            return [upsample_convex8(s_disp, s_mask)]
        outs = []
        for _ in range(iters):
            disp, hidden, mask = self.update(cost_volume, disp, left_image, right_image, hidden, True)
            outs.append(upsample_convex8(disp, mask))
        return outs
