"""Gaussian-kernel interpolation decoder (registry alternate 'interp'): mirror of the reference's
model/decoder/interpolation_decoder.py -- same constructor kwargs and state_dict keys."""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops
from .blocks import ResnetBlockFC


class PointInterpDecoder(nn.Module):
    """xyz_q [B,NQ,3] + encoding -> [B,NQ,out_dim]: anchor features blended with normalised Gaussian weights
    exp(-(|q - a| + 1e-5)^2 / 0.2^2), then the same conditioned ResNet MLP as the cross-attention decoder
    (reference model/decoder/interpolation_decoder.py:8-88)."""

    def __init__(self, dim_inp, dim, out_dim=3, hidden_dim=50, n_blocks=5):
        super().__init__()
        self.n_blocks = n_blocks
        self.fc0 = nn.Linear(dim_inp, dim)
        self.fc1 = nn.Linear(dim, hidden_dim)
        self.blocks = nn.ModuleList([ResnetBlockFC(hidden_dim) for _ in range(n_blocks)])
        self.fc_c = nn.ModuleList([nn.Linear(dim, hidden_dim) for _ in range(n_blocks)])
        self.fc_out = nn.Linear(hidden_dim, out_dim)
        self.var = 0.2 ** 2

    def sample_point_feature(self, q, p, fea):
        dist = -((p.unsqueeze(1) - q.unsqueeze(2)).norm(dim=3) + 10e-6) ** 2       # [B, NQ, n_anchors]
        weight = (dist / self.var).exp()
        weight = weight / weight.sum(dim=2, keepdim=True)
        return torch.bmm(weight, fea)

    def forward(self, xyz_q, encoding):
        lat = ops.linear(self.sample_point_feature(xyz_q, encoding["anchors"], encoding["anchor_feats"]), self.fc0)
        net = ops.linear(lat, self.fc1, relu_in=True)
        for i in range(self.n_blocks):
            net = ops.linear(lat, self.fc_c[i], residual=net)
            net = self.blocks[i](net)
        return ops.linear(net, self.fc_out, relu_in=True)
