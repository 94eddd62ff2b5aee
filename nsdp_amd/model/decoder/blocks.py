"""Decoder blocks (state-dict compatible mirror of the reference's model/decoder/blocks.py)."""
from __future__ import annotations

import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ... import precision


# compute units the prefetched position-encoding GEMM leaves to the encoder's chain (NSDP_DECODER_PREFETCH=1 only)
PREFETCH_RESERVE_CUS = int(os.environ.get("NSDP_PREFETCH_RESERVE_CUS", "0"))


class CrossTransformerBlock(nn.Module):
    """Cross attention from every query point to its `nneigh` nearest anchors plus one global token
    (reference model/decoder/blocks.py:12-95, separate_delta=True: delta evaluated twice with the same
    weights -- identical values, so it is evaluated once here and autograd sums the two uses)."""

    def __init__(self, dim_inp, dim, nneigh=7, reduce_dim=True, separate_delta=True):
        super().__init__()
        self.dim = dim
        self.nneigh = nneigh
        self.separate_delta = separate_delta
        self.fc_delta = nn.Sequential(nn.Linear(3, dim), nn.ReLU(), nn.Linear(dim, dim))
        self.fc_gamma = nn.Sequential(nn.Linear(dim, dim), nn.ReLU(), nn.Linear(dim, dim))
        self.w_k_global = nn.Linear(dim_inp, dim, bias=False)
        self.w_v_global = nn.Linear(dim_inp, dim, bias=False)
        self.w_qs = nn.Linear(dim_inp, dim, bias=False)
        self.w_ks = nn.Linear(dim_inp, dim, bias=False)
        self.w_vs = nn.Linear(dim_inp, dim, bias=False)
        if not reduce_dim:
            self.fc = nn.Linear(dim, dim_inp)
        self.reduce_dim = reduce_dim

    def prefetch(self, xyz_q, xyz, after=None):
        """The part of forward() that needs only the query points and the ANCHOR COORDINATES -- the anchor search, the relative
        coordinates and the position encoding delta(q_i - a_j) (a K = 4 layer and a 200 x 200 GEMM over B * NQ * 7 rows: ~1.8 ms
        of a B = 32 step) -- enqueued on a stream of its own behind ``after`` (the stream that produces ``xyz``) and behind the
        current stream, i.e. BESIDE the encoder's forward chain, whose 500- and 100-point levels leave most of the chip idle.
        The reference computes it after the encoder (model/decoder/blocks.py:49-52, :77-79).  forward(prefetched=) joins."""
        from ... import hip_linear
        hip_linear.refresh_weight_packs(xyz_q.device)      # (on THIS stream: see there)
        main = torch.cuda.current_stream(xyz_q.device)
        s = ops.prefetch_stream(xyz_q.device)
        s.wait_stream(main)
        if after is not None:
            s.wait_stream(after)
        with torch.cuda.stream(s):
            idx = ops.knn_indices(xyz_q, xyz, self.nneigh)
            rel = ops.relative_coords(xyz_q, xyz, idx)
            # (the position-encoding GEMM is a persistent full-chip kernel: it leaves PREFETCH_RESERVE_CUS compute units to the
            # encoder's chain, like the weight-gradient kernels of the backward pass do)
            hip_linear.lib().nsdp_debug_set(9, PREFETCH_RESERVE_CUS)
            try:
                pos = ops.mlp2(rel, self.fc_delta)
            finally:
                hip_linear.lib().nsdp_debug_set(9, 0)
        return {"xyz_q": xyz_q, "xyz": xyz, "idx": idx, "rel": rel, "pos": pos, "stream": s}

    def forward(self, xyz_q, lat_rep, xyz, points, prefetched=None, idx=None):
        """``idx`` [B,NQ,k] int32: the queries' anchor neighbours when the caller searched ahead of this pass
        (Deformation_Networks.geometry), else searched here."""
        assert lat_rep.dim() == 2, "per-query latent codes are not used by any NSDP configuration"
        pos = None
        if prefetched is not None and prefetched["xyz_q"] is xyz_q and prefetched["xyz"] is xyz:
            main = torch.cuda.current_stream(xyz_q.device)
            main.wait_stream(prefetched["stream"])
            idx, rel, pos = prefetched["idx"], prefetched["rel"], prefetched["pos"]
            for t in (idx, rel, pos):
                t.record_stream(main)
        elif idx is None:
            idx = ops.knn_indices(xyz_q, xyz, self.nneigh)                   # [B,NQ,k]
        q = ops.linear(lat_rep, self.w_qs)                                   # [B,D]  (shared by all queries)
        k_g = ops.linear(lat_rep, self.w_k_global)
        v_g = ops.linear(lat_rep, self.w_v_global)
        kf = ops.linear(points, self.w_ks)                                   # [B,A,D] anchor tables
        vf = ops.linear(points, self.w_vs)
        if pos is None:
            rel = ops.relative_coords(xyz_q, xyz, idx)                       # xyz_q - a_j
        logit_g = ops.mlp2(q - k_g, self.fc_gamma)                           # [B,D]: identical for all queries
        res, _ = ops.vector_attention(rel, q.unsqueeze(1), kf, vf, idx, self.fc_delta,
                                      self.fc_gamma, a_g=logit_g, v_g=v_g, pos=pos)
        if not self.reduce_dim:
            res = ops.linear(res, self.fc)
        return res


class ResnetBlockFC(nn.Module):
    """x + fc_1(relu(fc_0(relu(x))))   (reference model/decoder/blocks.py:99-142)."""

    def __init__(self, size_in, size_out=None, size_h=None):
        super().__init__()
        size_out = size_in if size_out is None else size_out
        size_h = min(size_in, size_out) if size_h is None else size_h
        self.size_in, self.size_h, self.size_out = size_in, size_h, size_out
        self.fc_0 = nn.Linear(size_in, size_h)
        self.fc_1 = nn.Linear(size_h, size_out)
        self.shortcut = None if size_in == size_out else nn.Linear(size_in, size_out, bias=False)
        nn.init.zeros_(self.fc_1.weight)

    def forward(self, x):
        x_s = x if self.shortcut is None else ops.linear(x, self.shortcut)
        if precision.is_bf16():        # (see ops.mlp2: the inner ReLU as fc_1's input ReLU -- no mask streams in backward)
            h = ops.linear(x, self.fc_0, relu_in=True)
            return ops.linear(h, self.fc_1, relu_in=True, residual=x_s)
        pair = ops.PAIR_MASK and torch.is_grad_enabled()                 # backward mask contract of ops.mlp2
        # identity shortcut: the skip connection's gradient joins fc_0's dX inside that GEMM (hip_linear.SkipGrad)
        skip = ops.skip_grad(x) if self.shortcut is None else None
        # (h is private to the two layers: G16 layout where the kernels have the form -- hip_linear.g16_pair_ok)
        g16 = not pair and ops.g16_pair(x, self.fc_0, self.fc_1, relu_in0=True)
        h = ops.linear(x, self.fc_0, relu_in=True, relu=True, premasked=pair, skip_dst=skip,
                       lay=ops.hip_linear.LAY_Y if g16 else 0)                                    # relu(fc_0(relu(x)))
        return ops.linear(h, self.fc_1, residual=x_s, mask_dx=pair, skip_src=skip,
                          lay=ops.hip_linear.LAY_X if g16 else 0)                                 # x_s + fc_1(h), fused epilogue
