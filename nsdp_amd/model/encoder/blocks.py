"""Encoder blocks of the Point-Transformer encoder on MI355X.

State-dict compatible with the reference's model/encoder/blocks.py (same attribute names, same
parameter shapes), so reference checkpoints load unchanged; the computation goes through
``nsdp_amd.model.ops`` (HIP kernels + channels-last layout) instead of materialised ATen tensors.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops


def _pair_mlp(d_in, d):
    return nn.Sequential(nn.Linear(d_in, d), nn.ReLU(), nn.Linear(d, d))


class TransformerBlock(nn.Module):
    """Local / global vector self-attention (reference model/encoder/blocks.py:52-134).

    y_i = BN( sum_j softmax_j[gamma(W_q x_i - W_k x_j + delta(xyz_i - xyz_j))] * (W_v x_j + delta) + x_i );
    ``pos_only``: logits = gamma(delta), values = delta, no residual (W_q/k/v exist but are unused).
    """

    def __init__(self, d_model, k, pos_only=False, group_all=False):
        super().__init__()
        self.pos_only = pos_only
        self.bn = nn.BatchNorm1d(d_model)
        self.fc_delta = _pair_mlp(3, d_model)
        self.fc_gamma = _pair_mlp(d_model, d_model)
        self.w_qs = nn.Linear(d_model, d_model, bias=False)
        self.w_ks = nn.Linear(d_model, d_model, bias=False)
        self.w_vs = nn.Linear(d_model, d_model, bias=False)
        self.k = k
        self.group_all = group_all

    def forward(self, xyz, feats=None, idx=None, inv=None, rel=None):
        """``idx`` [B,n,k] int32: precomputed neighbour indices (geometry pyramid); computed here if None.
        ``inv``: the inverse lists of ``idx`` when the pyramid built them as well.  ``rel``: xyz_i - xyz_j over ``idx`` when the
        caller has it already (blocks that share an index set share their relative coordinates)."""
        B, n, _ = xyz.shape
        if idx is not None:
            pass
        elif self.group_all:
            idx = torch.arange(n, device=xyz.device, dtype=torch.int32).view(1, 1, n).expand(B, n, n).contiguous()
        else:
            idx = ops.knn_indices(xyz, xyz, self.k)
        if rel is None:
            rel = ops.relative_coords(xyz, xyz, idx)                 # xyz_i - xyz_j
        if self.pos_only:
            res, _ = ops.vector_attention(rel, None, None, None, idx, self.fc_delta, self.fc_gamma)
        else:
            # the three projections read the same tensor: their input gradients are summed inside the dX GEMMs
            fan = ops.input_grad_sum(feats)
            q = ops.linear(feats, self.w_qs, grad_sum=fan)
            kf = ops.linear(feats, self.w_ks, grad_sum=fan)
            # where u = q - k + pos comes straight out of the position-encoding GEMM the values are rebuilt from u and the
            # table v + k: the value projection adds k itself (as its residual, a constant of that node)
            fused = ops.fused_pre_applies(idx, feats.shape[-1])
            vf = ops.linear(feats, self.w_vs, grad_sum=fan, residual=kf.detach() if fused else None)
            res, _ = ops.vector_attention(rel, q, kf, vf, idx, self.fc_delta, self.fc_gamma, residual=feats, combined=fused,
                                          inv=inv)
        return ops.batch_norm(res, self.bn)


class ElementwiseMLP(nn.Module):
    """bn3(x + relu(bn2(conv2(relu(bn1(conv1(x)))))))   (reference model/encoder/blocks.py:137-159)."""

    def __init__(self, dim):
        super().__init__()
        self.conv1 = nn.Conv1d(dim, dim, 1)
        self.bn1 = nn.BatchNorm1d(dim)
        self.conv2 = nn.Conv1d(dim, dim, 1)
        self.bn2 = nn.BatchNorm1d(dim)
        self.bn3 = nn.BatchNorm1d(dim)

    def forward(self, x):
        h = ops.batch_norm(ops.linear(x, self.conv1), self.bn1, relu=True)
        h = ops.batch_norm(ops.linear(h, self.conv2), self.bn2, relu=True)
        return ops.batch_norm(x, self.bn3, addend=h)                      # bn3(x + h), add fused


class TransformerSetAbstraction(nn.Module):
    """Attentive set abstraction: FPS down-sampling + two cross-attentions from each centre to its k
    nearest input points (reference model/encoder/blocks.py:220-314)."""

    def __init__(self, npoint, nneigh, dim):
        super().__init__()
        self.npoint = npoint
        self.nneigh = nneigh
        self.bnorm0 = nn.BatchNorm1d(dim)
        self.bnorm1 = nn.BatchNorm1d(dim)
        self.bnorm2 = nn.BatchNorm1d(dim)
        self.bn1 = nn.BatchNorm1d(dim)
        self.conv1 = nn.Conv1d(dim, dim, 1)
        self.conv2 = nn.Conv1d(dim, dim, 1)
        self.fc_delta1 = _pair_mlp(3, dim)
        self.fc_gamma1 = _pair_mlp(dim, dim)
        self.fc_gamma2 = _pair_mlp(dim, dim)
        self.w_qs = nn.Linear(dim, dim, bias=False)
        self.w_ks = nn.Linear(dim, dim, bias=False)
        self.w_vs = nn.Linear(dim, dim, bias=False)
        self.w_qs2 = nn.Linear(dim, dim, bias=False)
        self.w_ks2 = nn.Linear(dim, dim, bias=False)
        self.w_vs2 = nn.Linear(dim, dim, bias=False)

    def forward(self, xyz, points, geo=None):
        """``geo``: precomputed {fps_idx, new_xyz, sa_idx} of this level (geometry pyramid), else computed here."""
        if geo is not None:
            fps_idx, new_xyz, idx = geo["fps_idx"], geo["new_xyz"], geo["sa_idx"]
            inv = geo.get("sa_inv")
        else:
            inv = None
            fps_idx = ops.fps_indices(xyz, self.npoint)                   # [B, npoint] int32
            new_xyz = ops.index_points(xyz.detach(), fps_idx)             # detached centres (no_grad in ref)
            idx = ops.knn_indices(new_xyz, xyz, self.nneigh)              # [B, npoint, k]
        rel = ops.relative_coords(new_xyz, xyz, idx, sign=-1.0)           # xyz_j - c  (sign opposite to PTB)

        # the reference projects all N points with w_qs and then gathers the centres; gathering first
        # is the same values with N/npoint fewer rows through the GEMM
        q1 = ops.linear(ops.index_points(points, fps_idx), self.w_qs)
        fan = ops.input_grad_sum(points)          # four projections of `points`: one running sum through their dX GEMMs
        fused = ops.fused_pre_applies(idx, points.shape[-1])        # (see TransformerBlock: then the value table is v + k)
        k1 = ops.linear(points, self.w_ks, grad_sum=fan)
        v1 = ops.linear(points, self.w_vs, grad_sum=fan, residual=k1.detach() if fused else None)
        res1, pos = ops.vector_attention(rel, q1, k1, v1, idx, self.fc_delta1, self.fc_gamma1, combined=fused, inv=inv)
        res1 = ops.linear(ops.batch_norm(ops.linear(res1, self.conv1), self.bn1), self.conv2, relu_in=True,
                          residual=res1)
        res1 = ops.batch_norm(res1, self.bnorm0)

        # second attention re-uses pos; "res1 + res2" is fused as the kernel's residual add
        if isinstance(pos, ops.PosAsU):
            # pos exists only as u1 = q1 - k1 + pos: u2 = u1 + (q2 - q1) - (k2 - k1), values = u1 + (v2 + k1) - q1; the three
            # projections deliver those differences / sums themselves (signed residuals: constants of their nodes)
            q2 = ops.linear(res1, self.w_qs2, residual=pos.q, residual_sign=-1.0)
            k2 = ops.linear(points, self.w_ks2, grad_sum=fan, residual=pos.kf, residual_sign=-1.0)
            v2 = ops.linear(points, self.w_vs2, grad_sum=fan, residual=pos.kf)
            res12, _ = ops.vector_attention(None, q2, k2, v2, idx, None, self.fc_gamma2, residual=res1, pos=pos, combined=True,
                                            inv=inv)
        else:
            q2 = ops.linear(res1, self.w_qs2)
            res12, _ = ops.vector_attention(None, q2, ops.linear(points, self.w_ks2, grad_sum=fan),
                                            ops.linear(points, self.w_vs2, grad_sum=fan), idx,
                                            None, self.fc_gamma2, residual=res1, pos=pos, inv=inv)

        new_points = ops.batch_norm(res12, self.bnorm1)
        return new_xyz, ops.batch_norm(new_points, self.bnorm2, addend=ops.index_points(points, fps_idx))


class PointNetSetAbstraction(nn.Module):
    """PointNet++-style set abstraction: FPS, per-point residual MLP, max over the k nearest input points
    (registry alternate, reference model/encoder/blocks.py:162-217)."""

    def __init__(self, npoint, nneigh, in_channel, dim):
        super().__init__()
        self.npoint = npoint
        self.nneigh = nneigh
        self.fc1 = nn.Linear(in_channel, dim)
        self.conv1 = nn.Conv1d(dim, dim, 1)
        self.conv2 = nn.Conv1d(dim, dim, 1)
        self.bn1 = nn.BatchNorm1d(dim)
        self.bn2 = nn.BatchNorm1d(dim)
        self.bn = nn.BatchNorm1d(dim)

    def forward(self, xyz, points):
        fps_idx = ops.fps_indices(xyz, self.npoint)
        new_xyz = ops.index_points(xyz, fps_idx)
        points = ops.linear(points, self.fc1)
        points_ori = ops.index_points(points, fps_idx)
        h = ops.batch_norm(ops.linear(points, self.conv1), self.bn1, relu=True)
        points = points + ops.batch_norm(ops.linear(h, self.conv2), self.bn2, relu=True)
        idx = ops.knn_indices(new_xyz, xyz, self.nneigh)
        new_points = points_ori + ops.index_points(points, idx).max(dim=2)[0]
        return new_xyz, ops.batch_norm(new_points, self.bn)


class TransitionDown(nn.Module):
    """Wrapper selecting the set-abstraction flavour (reference model/encoder/blocks.py:18-49)."""

    def __init__(self, npoint, nneighbor, dim, type="attentive"):
        super().__init__()
        if type == "attentive":
            self.sa = TransformerSetAbstraction(npoint, nneighbor, dim)
        elif type == "maxpool":
            self.sa = PointNetSetAbstraction(npoint, nneighbor, dim, dim)
        else:
            raise ValueError("Set Abstraction type " + type + " unknown!")

    def forward(self, xyz, feats, geo=None):
        return self.sa(xyz, feats, geo) if geo is not None else self.sa(xyz, feats)
