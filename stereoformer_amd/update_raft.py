"""RAFT-Stereo multi-level ConvGRU update operator (A14) behind
``models/IterativeLocalCostVolume/update/update.py``.

Convolutions / pooling stay with the host framework (MIOpen); the gate math
(``sigmoid(conv+c)*h``, ``(1-z)h + z*tanh(conv+c)``) and ``interp`` are HIP kernels.
Parameter names match the reference so its ``state_dict``s load unchanged.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from .update_dlc import gru_blend, gru_reset
from .upsample import interp


class FlowHead(nn.Module):
    """``FlowHead`` (update/update.py:6-14)."""

    def __init__(self, input_dim=128, hidden_dim=256, output_dim=2):
        super().__init__()
        self.conv1 = nn.Conv2d(input_dim, hidden_dim, 3, padding=1)
        self.conv2 = nn.Conv2d(hidden_dim, output_dim, 3, padding=1)
        self.relu = nn.ReLU(inplace=True)

    def forward(self, x):
        return self.conv2(self.relu(self.conv1(x)))


class ConvGRU(nn.Module):
    """``ConvGRU(hidden_dim, input_dim)`` with precomputed context biases (update/update.py:16-33)."""

    def __init__(self, hidden_dim, input_dim, kernel_size=3):
        super().__init__()
        self.convz = nn.Conv2d(hidden_dim + input_dim, hidden_dim, kernel_size, padding=kernel_size // 2)
        self.convr = nn.Conv2d(hidden_dim + input_dim, hidden_dim, kernel_size, padding=kernel_size // 2)
        self.convq = nn.Conv2d(hidden_dim + input_dim, hidden_dim, kernel_size, padding=kernel_size // 2)

    def forward(self, h, cz, cr, cq, *x_list):
        x = torch.cat(x_list, dim=1)
        hx = torch.cat([h, x], dim=1)
        zc = self.convz(hx)
        rh = gru_reset(h, self.convr(hx), cr)                       # sigmoid(convr + cr) * h
        qc = self.convq(torch.cat([rh, x], dim=1))
        return gru_blend(h, zc, qc, cz, cq)                         # (1-z)h + z tanh(convq + cq)


class SepConvGRU(nn.Module):
    """``SepConvGRU`` (update/update.py:35-63): horizontal (1x5) then vertical (5x1) GRU."""

    def __init__(self, hidden_dim=128, input_dim=192 + 128):
        super().__init__()
        self.convz1 = nn.Conv2d(hidden_dim + input_dim, hidden_dim, (1, 5), padding=(0, 2))
        self.convr1 = nn.Conv2d(hidden_dim + input_dim, hidden_dim, (1, 5), padding=(0, 2))
        self.convq1 = nn.Conv2d(hidden_dim + input_dim, hidden_dim, (1, 5), padding=(0, 2))
        self.convz2 = nn.Conv2d(hidden_dim + input_dim, hidden_dim, (5, 1), padding=(2, 0))
        self.convr2 = nn.Conv2d(hidden_dim + input_dim, hidden_dim, (5, 1), padding=(2, 0))
        self.convq2 = nn.Conv2d(hidden_dim + input_dim, hidden_dim, (5, 1), padding=(2, 0))

    def forward(self, h, *x):
        x = torch.cat(x, dim=1)
        for cz, cr, cq in ((self.convz1, self.convr1, self.convq1), (self.convz2, self.convr2, self.convq2)):
            hx = torch.cat([h, x], dim=1)
            zc = cz(hx)
            rh = gru_reset(h, cr(hx))
            h = gru_blend(h, zc, cq(torch.cat([rh, x], dim=1)))
        return h


class BasicMotionEncoder_LZH(nn.Module):
    """``BasicMotionEncoder_LZH`` (update/update.py:66-91)."""

    def __init__(self, corr_levels, corr_radius):
        super().__init__()
        self.corr_levels = corr_levels
        self.corr_radius = corr_radius
        cor_planes = self.corr_levels * (2 * self.corr_radius + 1)
        self.convc1 = nn.Conv2d(cor_planes, 64, 1, padding=0)
        self.convc2 = nn.Conv2d(64, 64, 3, padding=1)
        self.convf1 = nn.Conv2d(2, 64, 7, padding=3)
        self.convf2 = nn.Conv2d(64, 64, 3, padding=1)
        self.conv = nn.Conv2d(64 + 64, 128 - 2, 3, padding=1)

    def forward(self, flow, corr):
        cor = F.relu(self.convc1(corr))
        cor = F.relu(self.convc2(cor))
        flo = F.relu(self.convf1(flow))
        flo = F.relu(self.convf2(flo))
        out = F.relu(self.conv(torch.cat([cor, flo], dim=1)))
        return torch.cat([out, flow], dim=1)


def pool2x(x):
    """``pool2x`` (update/update.py:119-120)."""
    return F.avg_pool2d(x, 3, stride=2, padding=1)


def pool4x(x):
    """``pool4x`` (update/update.py:122-123)."""
    return F.avg_pool2d(x, 5, stride=4, padding=1)


class BasicMultiUpdateBlock_LZH(nn.Module):
    """``BasicMultiUpdateBlock_LZH(corr_levels, corr_radius, n_gru_layers, n_downsample, hidden_dims)``
    (update/update.py:130-178).  Mutates ``net`` in place like the reference."""

    def __init__(self, corr_levels, corr_radius, n_gru_layers, n_downsample, hidden_dims=[]):
        super().__init__()
        self.n_gru_layers = n_gru_layers
        self.n_downsample = n_downsample
        self.encoder = BasicMotionEncoder_LZH(corr_levels=corr_levels, corr_radius=corr_radius)
        encoder_output_dim = 128
        self.gru08 = ConvGRU(hidden_dims[2], encoder_output_dim + hidden_dims[1] * (self.n_gru_layers > 1))
        self.gru16 = ConvGRU(hidden_dims[1], hidden_dims[0] * (self.n_gru_layers == 3) + hidden_dims[2])
        self.gru32 = ConvGRU(hidden_dims[0], hidden_dims[1])
        self.flow_head = FlowHead(hidden_dims[2], hidden_dim=256, output_dim=2)
        factor = 2 ** self.n_downsample
        self.mask = nn.Sequential(
            nn.Conv2d(hidden_dims[2], 256, 3, padding=1), nn.ReLU(inplace=True),
            nn.Conv2d(256, (factor ** 2) * 9, 1, padding=0))

    def forward(self, net, inp, corr=None, flow=None, iter08=True, iter16=True, iter32=True, update=True):
        with torch.no_grad():
            if iter32:
                net[2] = self.gru32(net[2], *(inp[2]), pool2x(net[1]))
            if iter16:
                if self.n_gru_layers > 2:
                    net[1] = self.gru16(net[1], *(inp[1]), pool2x(net[0]), interp(net[2], net[1]))
                else:
                    net[1] = self.gru16(net[1], *(inp[1]), pool2x(net[0]))
            if iter08:
                motion_features = self.encoder(flow, corr)
                if self.n_gru_layers > 1:
                    net[0] = self.gru08(net[0], *(inp[0]), motion_features, interp(net[1], net[0]))
                else:
                    net[0] = self.gru08(net[0], *(inp[0]), motion_features)
            if not update:
                return net
            delta_flow = self.flow_head(net[0])
            mask = .25 * self.mask(net[0])  # scale mask to balance gradients (:176)
        return net, mask, delta_flow
