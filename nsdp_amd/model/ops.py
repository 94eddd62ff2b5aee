"""Functional building blocks of the TDNet hot path on MI355X.

Feature tensors are channels-last ``[B, n, C]`` everywhere (the reference permutes to ``[B, C, n]``
around every BatchNorm1d / Conv1d, e.g. model/encoder/blocks.py:132,:158; per-channel statistics over
B*n rows are identical).  Index tensors are int32 ``[B, n, k]`` produced by the HIP kNN / FPS kernels.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import hip_linear
from .. import pointnet2_utils as pu


# ---------------------------------------------------------------------------------------------
# geometry (hand-written HIP, non-differentiable exactly like the reference's no_grad blocks)
# ---------------------------------------------------------------------------------------------
@torch.no_grad()
def knn_indices(query: torch.Tensor, source: torch.Tensor, k: int) -> torch.Tensor:
    return pu.knn(query.detach().contiguous(), source.detach().contiguous(), k)


@torch.no_grad()
def fps_indices(xyz: torch.Tensor, npoint: int) -> torch.Tensor:
    return pu.furthest_point_sample(xyz.detach().contiguous(), npoint)


class _GatherRows(torch.autograd.Function):
    """index_points (model/utils.py:58-70) for idx [B,S] or [B,S,K]; backward = scatter-add."""

    @staticmethod
    def forward(ctx, points, idx):
        B, N, C = points.shape
        flat = idx.reshape(B, -1)
        ctx.save_for_backward(flat)
        ctx.n = N
        out = pu.gather_rows(points.contiguous(), flat)
        return out.reshape(*idx.shape, C)

    @staticmethod
    def backward(ctx, grad_out):
        (flat,) = ctx.saved_tensors
        B = flat.shape[0]
        g = grad_out.reshape(B, flat.shape[1], -1).contiguous()
        return pu.scatter_add_rows(g, flat, ctx.n), None


def index_points(points: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    return _GatherRows.apply(points, idx)


# ---------------------------------------------------------------------------------------------
# dense layers
# ---------------------------------------------------------------------------------------------
def linear(x: torch.Tensor, lin, relu: bool = False, relu_in: bool = False, residual=None) -> torch.Tensor:
    """nn.Linear / 1x1 nn.Conv1d on channels-last rows, on the fp32 matrix cores (hip_linear):
    relu?( relu_in?(x) @ W^T + b (+ residual) )."""
    return hip_linear.linear(x, lin.weight, lin.bias, relu_in=relu_in, relu_out=relu, residual=residual)


def mlp2(x: torch.Tensor, seq: nn.Sequential) -> torch.Tensor:
    """nn.Sequential(Linear, ReLU, Linear) (fc_delta / fc_gamma / fc_middle)."""
    return linear(linear(x, seq[0], relu=True), seq[2])


def batch_norm(x: torch.Tensor, bn: nn.BatchNorm1d) -> torch.Tensor:
    """BatchNorm1d over (B*n) rows per channel; batch statistics + running update when training."""
    shape = x.shape
    if bn.training and bn.track_running_stats:
        bn.num_batches_tracked.add_(1)
    y = F.batch_norm(x.reshape(-1, shape[-1]), bn.running_mean, bn.running_var, bn.weight, bn.bias,
                     bn.training, bn.momentum, bn.eps)
    return y.reshape(shape)


# ---------------------------------------------------------------------------------------------
# point-transformer vector attention
# ---------------------------------------------------------------------------------------------
def vector_attention(rel: torch.Tensor, q, k_nb, v_nb, fc_delta, fc_gamma):
    """softmax over the neighbour axis, independently per channel, of gamma(q - k + delta(rel)),
    applied to (v + delta(rel)).  rel [B,n,k,3]; q [B,n,d] or None (pos_only); k_nb, v_nb [B,n,k,d].
    Returns (aggregate [B,n,d], pos_encode [B,n,k,d])."""
    pos = mlp2(rel, fc_delta)
    if q is None:
        logits = mlp2(pos, fc_gamma)
        val = pos
    else:
        logits = mlp2(q.unsqueeze(2) - k_nb + pos, fc_gamma)
        val = v_nb + pos
    w = F.softmax(logits, dim=-2)
    return (w * val).sum(dim=2), pos


def attention_with_pos(pos, q, k_nb, v_nb, fc_gamma):
    """Second attention of TransformerSetAbstraction: re-uses an already computed pos_encode."""
    w = F.softmax(mlp2(q.unsqueeze(2) - k_nb + pos, fc_gamma), dim=-2)
    return (w * (v_nb + pos)).sum(dim=2)
