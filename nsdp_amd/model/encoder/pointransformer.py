"""Point-Transformer encoder (mirror of the reference's model/encoder/pointransformer.py)."""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops
from .blocks import ElementwiseMLP, TransformerBlock, TransitionDown


class PointTransformerEncoder(nn.Module):
    """input [B,N,3(+F)] -> {'z': [B,d], 'anchors': [B,n_last,3], 'anchor_feats': [B,n_last,d]}
    (reference model/encoder/pointransformer.py:27-140; same constructor kwargs and state_dict keys)."""

    def __init__(self, npoints_per_layer, nneighbor, nneighbor_reduced, nfinal_transformers,
                 d_transformer, d_reduced, full_SA=False, has_features=False, inp_feat_dim=1):
        super().__init__()
        self.d_reduced = d_reduced
        self.d_transformer = d_transformer
        self.has_features = has_features
        self.fc_middle = nn.Sequential(nn.Linear(d_transformer, d_transformer), nn.ReLU(),
                                       nn.Linear(d_transformer, d_transformer))
        if has_features:
            self.enc_sdf = nn.Linear(inp_feat_dim, d_reduced)
        self.transformer_begin = TransformerBlock(d_reduced, nneighbor_reduced, pos_only=not has_features)
        self.transition_downs = nn.ModuleList()
        self.transformer_downs = nn.ModuleList()
        self.elementwise = nn.ModuleList()
        self.elementwise_extras = nn.ModuleList()
        if d_reduced != d_transformer:
            self.fc1 = nn.Linear(d_reduced, d_transformer)
        for i in range(len(npoints_per_layer) - 1):
            old_n, new_n = npoints_per_layer[i], npoints_per_layer[i + 1]
            dim = d_reduced if i == 0 else d_transformer
            self.transition_downs.append(TransitionDown(new_n, min(nneighbor, old_n), dim))
            self.elementwise_extras.append(ElementwiseMLP(dim))
            self.transformer_downs.append(TransformerBlock(dim, min(nneighbor, new_n)))
            self.elementwise.append(ElementwiseMLP(d_transformer))
        self.final_transformers = nn.ModuleList(
            [TransformerBlock(d_transformer, 2 * nneighbor, group_all=full_SA) for _ in range(nfinal_transformers)])
        self.final_elementwise = nn.ModuleList(
            [ElementwiseMLP(dim=d_transformer) for _ in range(nfinal_transformers)])

    def forward(self, xyz, on_anchors=None):
        """``on_anchors(anchors, after)``: called as soon as the anchor coordinates (the last level of the geometry pyramid) are
        ENQUEUED -- on the pyramid's stream ``after``, ~1 ms into a step -- so that work which needs nothing else of the encoding
        (the decoder's anchor search and position encoding, CrossTransformerDecoder.prefetch) can be launched beside the
        encoder's forward chain.  Whatever it returns travels in the encoding as ``'prefetch'``."""
        coords = xyz[:, :, :3].contiguous() if self.has_features else xyz
        # all FPS / kNN index tensors of the pyramid, launched on a side stream under transformer_begin
        levels, join = ops.geometry_pyramid(
            coords, [td.sa.npoint for td in self.transition_downs],
            [(td.sa.nneigh, None if tb.group_all else tb.k)
             for td, tb in zip(self.transition_downs, self.transformer_downs)],
            dims=([self.d_reduced] + [self.d_transformer] * (len(self.transition_downs) - 1)) if torch.is_grad_enabled() else None)
        prefetch = on_anchors(levels[-1]["new_xyz"], ops.geometry_stream(coords.device)) if (on_anchors and levels) else None
        if self.has_features:
            feats = ops.linear(xyz[:, :, 3:], self.enc_sdf)
            xyz = coords
            feats = self.transformer_begin(xyz, feats)
        else:
            feats = self.transformer_begin(xyz)
        join()
        for i in range(len(self.transition_downs)):
            xyz, feats = self.transition_downs[i](xyz, feats, levels[i])
            feats = self.elementwise_extras[i](feats)
            feats = self.transformer_downs[i](xyz, feats, idx=levels[i]["blk_idx"], inv=levels[i].get("blk_inv"))
            if i == 0 and self.d_reduced != self.d_transformer:
                feats = ops.linear(feats, self.fc1)
            feats = self.elementwise[i](feats)
        final_idx = None      # the final blocks all search the same cloud with the same k: one kNN (and one inverse list) for all
        for blk, mlp in zip(self.final_transformers, self.final_elementwise):
            if final_idx is None and not blk.group_all:
                final_idx = ops.knn_indices(xyz, xyz, blk.k)
            feats = mlp(blk(xyz, feats, idx=final_idx))
        lat_vec = feats.max(dim=1)[0]
        enc = {"z": ops.mlp2(lat_vec, self.fc_middle), "anchors": xyz, "anchor_feats": feats}
        if prefetch is not None:
            enc["prefetch"] = prefetch
        return enc
