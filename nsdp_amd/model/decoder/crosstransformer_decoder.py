"""Cross-attention decoder (mirror of the reference's model/decoder/crosstransformer_decoder.py)."""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops
from ... import hip_decoder, hip_linear, precision
from .blocks import CrossTransformerBlock, ResnetBlockFC

# bf16 storage: the decoder's residual trunk in fp32 storage (NSDP_BF16_TRUNK=f32), see forward()
TRUNK_F32 = __import__("os").environ.get("NSDP_BF16_TRUNK", "bf16") == "f32"


class CrossTransformerDecoder(nn.Module):
    """xyz_q [B,NQ,3] + encoding -> [B,NQ,out_dim]
    (reference model/decoder/crosstransformer_decoder.py:24-70)."""

    def __init__(self, dim_inp, dim, nneigh=7, hidden_dim=64, n_blocks=5, out_dim=1):
        super().__init__()
        self.dim = dim
        self.n_blocks = n_blocks
        self.ct1 = CrossTransformerBlock(dim_inp, dim, nneigh=nneigh)
        self.init_enc = nn.Linear(dim, hidden_dim)
        self.blocks = nn.ModuleList([ResnetBlockFC(hidden_dim) for _ in range(n_blocks)])
        self.fc_c = nn.ModuleList([nn.Linear(dim, hidden_dim) for _ in range(n_blocks)])
        self.fc_out = nn.Linear(hidden_dim, out_dim)

    def prefetch(self, xyz_q, anchors, after=None):
        """CrossTransformerBlock.prefetch for this decoder's attention block (None where forward() would not use it: the
        no-grad path runs the fused whole-decoder kernel)."""
        if (hip_decoder.ENABLED and not torch.is_grad_enabled() and hip_decoder.supported(self) and not precision.is_bf16()):
            return None
        return self.ct1.prefetch(xyz_q, anchors, after)

    @staticmethod
    def _query_idx(xyz_q, encoding):
        """The anchor neighbours searched ahead of the pass (encoding['query_idx'], Deformation_Networks.geometry) if they are
        these queries' (encoding['query_points'] is xyz_q)."""
        return encoding.get("query_idx") if encoding.get("query_points") is xyz_q else None

    @torch.no_grad()
    def geometry(self, xyz_q, anchors):
        """The index set forward() derives from coordinates alone: each query's nearest anchors."""
        return {"query_idx": ops.knn_indices(xyz_q, anchors, self.ct1.nneigh)}

    def forward(self, xyz_q, encoding):
        if (hip_decoder.ENABLED and not torch.is_grad_enabled() and hip_decoder.supported(self)
                and not precision.is_bf16()):
            # inference: kNN + one fused kernel (18 dense layers + softmax in registers), nsdp_decoder_fused_fwd
            return hip_decoder.decoder_forward(self, xyz_q, encoding)
        lat = self.ct1(xyz_q, encoding["z"], encoding["anchors"], encoding["anchor_feats"], prefetched=encoding.get("prefetch"),
                       idx=self._query_idx(xyz_q, encoding))
        if precision.is_bf16() and TRUNK_F32:
            # bf16 storage keeps the [B, NQ, 7, 200] tensors of the attention block in bf16 -- 7 x the rows and 1.6 x the width
            # of the trunk -- while the residual stream `net` (128 wide, one row per query: the tensor that accumulates six
            # additions and becomes the output POSITION) stays fp32: a bf16 `net` rounds a coordinate-sized value to 8 bits
            # six times (tools/bf16_bisect.py)
            with precision.storage(torch.float32):
                return self._trunk(lat.float())
        return self._trunk(lat)

    def _trunk(self, lat):
        # the latent code feeds n_blocks + 1 layers: its gradient is summed inside their dX GEMMs
        fan = hip_linear.InputGradSum() if (torch.is_grad_enabled() and lat.requires_grad) else None
        net = ops.linear(lat, self.init_enc, grad_sum=fan)
        for i in range(self.n_blocks):
            net = ops.linear(lat, self.fc_c[i], residual=net, grad_sum=fan)   # net + fc_c[i](lat)
            net = self.blocks[i](net)
        return ops.linear(net, self.fc_out, relu_in=True, out_f32=True)   # fc_out(relu(net)); fp32 in every storage mode
