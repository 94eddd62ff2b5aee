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

    def _pyramid_args(self):
        return ([td.sa.npoint for td in self.transition_downs],
                [(td.sa.nneigh, None if tb.group_all else tb.k) for td, tb in zip(self.transition_downs, self.transformer_downs)],
                [self.d_reduced] + [self.d_transformer] * (len(self.transition_downs) - 1))

    @torch.no_grad()
    def geometry(self, xyz, training=None):
        """Every index set forward() derives from the COORDINATES alone -- the first block's neighbours, the sampling / grouping
        pyramid, the final blocks' neighbours and (``training``: default = grad mode) the inverse lists their backward passes
        scatter through -- computed on the current stream, as a dict forward(geometry=) takes instead of searching itself.
        Nothing in it depends on a parameter: a host that knows the NEXT batch computes it beside the current step
        (nsdp_amd.graph_step.PipelinedGeometry); the reference searches inside every forward (model/encoder/blocks.py:97-99,
        :283-288)."""
        training = torch.is_grad_enabled() if training is None else bool(training)
        coords = (xyz[:, :, :3] if self.has_features else xyz).detach().contiguous()
        npoints, ks, dims = self._pyramid_args()
        g = {}
        tb = self.transformer_begin
        if not tb.group_all:
            g["begin_idx"] = ops.knn_indices(coords, coords, tb.k)
            if training and not tb.pos_only:
                g["begin_inv"] = ops.attention_lists(g["begin_idx"], coords.shape[1], coords.shape[1], self.d_reduced)
        levels, _ = ops.geometry_pyramid(coords, npoints, ks, overlap=False, dims=dims if training else None)
        g["levels"] = levels
        last = levels[-1]["new_xyz"] if levels else coords
        blk = next((b for b in self.final_transformers if not b.group_all), None)
        if blk is not None:
            g["final_idx"] = ops.knn_indices(last, last, blk.k)
            if training:
                g["final_inv"] = ops.attention_lists(g["final_idx"], last.shape[1], last.shape[1], self.d_transformer)
        g["anchors"] = last
        return g

    def forward(self, xyz, on_anchors=None, geometry=None):
        """``on_anchors(anchors, after)``: called as soon as the anchor coordinates (the last level of the geometry pyramid) are
        ENQUEUED -- on the pyramid's stream ``after``, ~1 ms into a step -- so that work which needs nothing else of the encoding
        (the decoder's anchor search and position encoding, CrossTransformerDecoder.prefetch) can be launched beside the
        encoder's forward chain.  Whatever it returns travels in the encoding as ``'prefetch'``."""
        coords = xyz[:, :, :3].contiguous() if self.has_features else xyz
        # all FPS / kNN index tensors of the pyramid, launched on a side stream under transformer_begin
        begin = {}
        if geometry is not None:      # (geometry(): the index sets were computed ahead of this pass)
            levels, join = geometry["levels"], (lambda: None)
            begin = {"idx": geometry.get("begin_idx"), "inv": geometry.get("begin_inv")}
        else:
            npoints, ks, dims = self._pyramid_args()
            levels, join = ops.geometry_pyramid(coords, npoints, ks, dims=dims if torch.is_grad_enabled() else None)
        prefetch = (on_anchors(levels[-1]["new_xyz"], ops.geometry_stream(coords.device))
                    if (on_anchors and levels and geometry is None) else None)
        if self.has_features:
            feats = ops.linear(xyz[:, :, 3:], self.enc_sdf)
            xyz = coords
            feats = self.transformer_begin(xyz, feats, **begin)
        else:
            feats = self.transformer_begin(xyz, **begin)
        join()
        for i in range(len(self.transition_downs)):
            xyz, feats = self.transition_downs[i](xyz, feats, levels[i])
            feats = self.elementwise_extras[i](feats)
            feats = self.transformer_downs[i](xyz, feats, idx=levels[i]["blk_idx"], inv=levels[i].get("blk_inv"))
            if i == 0 and self.d_reduced != self.d_transformer:
                feats = ops.linear(feats, self.fc1)
            feats = self.elementwise[i](feats)
        # the final blocks all search the same cloud with the same k: one kNN (and one inverse list) for all
        final_idx = geometry.get("final_idx") if geometry is not None else None
        final_inv = geometry.get("final_inv") if geometry is not None else None
        # (... also when they attend to the whole cloud, group_all: one arange index tensor and one list instead of one per block --
        # a list build is a launch on the forward chain, where a kernel's microsecond is a step's microsecond)
        same = len({(blk.group_all, blk.k) for blk in self.final_transformers}) == 1
        final_rel = None
        for blk, mlp in zip(self.final_transformers, self.final_elementwise):
            if final_idx is None and (same or not blk.group_all):
                n = xyz.shape[1]
                if blk.group_all:
                    final_idx = torch.arange(n, device=xyz.device, dtype=torch.int32).view(1, 1, n).expand(xyz.shape[0], n, n).contiguous()
                else:
                    final_idx = ops.knn_indices(xyz, xyz, blk.k)
                if final_inv is None and same and not blk.pos_only:
                    final_inv = ops.hip_attention.backward_lists(final_idx, n, n, feats.shape[-1])
            shared = final_idx is not None and (same or not blk.group_all)
            if shared and final_rel is None and not xyz.requires_grad:      # (coordinates without a gradient: a constant of the blocks)
                final_rel = ops.relative_coords(xyz, xyz, final_idx)
            feats = mlp(blk(xyz, feats, idx=final_idx if shared else None, inv=final_inv if shared else None,
                            rel=final_rel if shared else None))
        lat_vec = feats.max(dim=1)[0]
        enc = {"z": ops.mlp2(lat_vec, self.fc_middle), "anchors": xyz, "anchor_feats": feats}
        if prefetch is not None:
            enc["prefetch"] = prefetch
        return enc
