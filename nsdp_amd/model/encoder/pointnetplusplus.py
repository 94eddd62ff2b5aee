"""PointNet++-style encoder (registry alternate 'pointnet++', used by the reference only in ablations): mirror of
model/encoder/pointnetplusplus.py -- same constructor kwargs, state_dict keys and output dict."""
from __future__ import annotations

import torch.nn as nn

from .. import ops
from .blocks import ElementwiseMLP, TransformerBlock, TransitionDown


class PointNetPlusPlusEncoder(nn.Module):
    """input [B,N,3(+F)] -> {'z': [B,d], 'anchors': [B,n_last,3], 'anchor_feats': [B,n_last,d]}
    (reference model/encoder/pointnetplusplus.py:5-96)."""

    def __init__(self, npoints_per_layer, nneighbor, d_transformer, nfinal_transformers, has_features=False,
                 inp_feat_dim=1):
        super().__init__()
        self.d_transformer = d_transformer
        self.has_features = has_features
        self.inp_feat_dim = inp_feat_dim
        self.fc_middle = nn.Sequential(nn.Linear(d_transformer, d_transformer), nn.ReLU(),
                                       nn.Linear(d_transformer, d_transformer))
        self.fc_begin = nn.Sequential(nn.Linear(inp_feat_dim if has_features else 3, d_transformer), nn.ReLU(),
                                      nn.Linear(d_transformer, d_transformer))
        self.transition_downs = nn.ModuleList()
        self.elementwise = nn.ModuleList()
        for i in range(len(npoints_per_layer) - 1):
            old_n, new_n = npoints_per_layer[i], npoints_per_layer[i + 1]
            self.transition_downs.append(TransitionDown(new_n, min(nneighbor, old_n), d_transformer, type="maxpool"))
            self.elementwise.append(ElementwiseMLP(d_transformer))
        self.final_transformers = nn.ModuleList(
            [TransformerBlock(d_transformer, -1, group_all=True) for _ in range(nfinal_transformers)])
        self.final_elementwise = nn.ModuleList([ElementwiseMLP(dim=d_transformer) for _ in range(nfinal_transformers)])

    def forward(self, xyz):
        if self.has_features:
            feats = ops.mlp2(xyz[:, :, 3:].contiguous(), self.fc_begin)
            xyz = xyz[:, :, 0:3].contiguous()
        else:
            feats = ops.mlp2(xyz, self.fc_begin)
        for i in range(len(self.transition_downs)):
            xyz, feats = self.transition_downs[i](xyz, feats)
            feats = self.elementwise[i](feats)
        for i, block in enumerate(self.final_transformers):
            feats = self.final_elementwise[i](block(xyz, feats))
        return {"z": ops.mlp2(feats.max(dim=1)[0], self.fc_middle), "anchors": xyz, "anchor_feats": feats}
