"""Functional building blocks of the TDNet hot path on MI355X.

Feature tensors are channels-last ``[B, n, C]`` everywhere (the reference permutes to ``[B, C, n]``
around every BatchNorm1d / Conv1d, e.g. model/encoder/blocks.py:132,:158; per-channel statistics over
B*n rows are identical).  Index tensors are int32 ``[B, n, k]`` produced by the HIP kNN / FPS kernels.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import hip_attention, hip_batchnorm, hip_linear, precision
from .. import pointnet2_utils as pu


# ---------------------------------------------------------------------------------------------
# geometry (hand-written HIP, non-differentiable exactly like the reference's no_grad blocks)
# ---------------------------------------------------------------------------------------------
# ABLATION, timing only (NSDP_GEOMETRY_ABLATE=1): every index set is computed once per (input address, shape) and reused -- what a
# step costs when no search, sampling or list build is on its chain (the ceiling of pipelining the next batch's geometry under
# the current step).  Results are those of the first batch: never for training.
_GEOMETRY_ABLATE = os.environ.get("NSDP_GEOMETRY_ABLATE", "0") == "1"
_ablate_cache = {}


@torch.no_grad()
def knn_indices(query: torch.Tensor, source: torch.Tensor, k: int) -> torch.Tensor:
    if _GEOMETRY_ABLATE:
        key = ("knn", tuple(query.shape), tuple(source.shape), k)      # (by shape: the bench feeds the same batch)
        if key not in _ablate_cache:
            _ablate_cache[key] = pu.knn(query.detach().contiguous(), source.detach().contiguous(), k)
        return _ablate_cache[key]
    return pu.knn(query.detach().contiguous(), source.detach().contiguous(), k)


@torch.no_grad()
def fps_indices(xyz: torch.Tensor, npoint: int) -> torch.Tensor:
    return pu.furthest_point_sample(xyz.detach().contiguous(), npoint)


def attention_lists(idx: torch.Tensor, n: int, N: int, d: int, per_shape_query: bool = False):
    """The inverse neighbour lists the BACKWARD pass of a d-wide attention block over ``idx`` [B,n,k] (sources: N) scatters
    through (hip_attention._use_inverse decides whether it uses any), or None -- for callers that prepare a block's geometry
    outside its forward pass, where grad mode says nothing."""
    if not hip_attention._use_inverse(torch.float32, per_shape_query, n, N, d):
        return None
    return hip_attention.inverse_lists(idx, N)


_side_streams = {}
# inverse neighbour lists of the pyramid's index sets built on the geometry stream (A/B knob: 0 = by the attention blocks, on
# the forward chain)
PYRAMID_LISTS = os.environ.get("NSDP_PYRAMID_LISTS", "1") != "0"


def _side_stream(device):
    key = (device.type, device.index)
    if key not in _side_streams:
        _side_streams[key] = torch.cuda.Stream(device=device)
    return _side_streams[key]


def geometry_stream(device):
    """The stream geometry_pyramid enqueues on (None when there is none yet)."""
    return _side_streams.get((device.type, device.index))


_prefetch_streams = {}


def prefetch_stream(device):
    key = (device.type, device.index)
    if key not in _prefetch_streams:
        _prefetch_streams[key] = torch.cuda.Stream(device=device)
    return _prefetch_streams[key]


@torch.no_grad()
def geometry_pyramid(xyz: torch.Tensor, npoints, ks, overlap: bool = True, dims=None):
    """Every index tensor of the encoder's down-sampling pyramid depends on coordinates only:
    FPS level 1 -> centres -> FPS level 2 -> centres, and the kNN sets of the set abstractions and of the
    local attention blocks at each level.  They are produced here in one go on a SIDE STREAM, so the ~600
    dependent FPS iterations and the kNN scans run underneath the first attention block's dense layers
    instead of in front of every later block (the reference serialises them, model/encoder/blocks.py:283-288).

    xyz [B,N,3]; npoints = [n1, n2, ...]; ks = [(k_sa, k_block), ...] per level.
    ``dims`` = the feature width of each level's attention blocks: where their backward pass will scatter through inverse
    neighbour lists (hip_attention._use_inverse) the lists are built HERE as well -- they depend on the index sets only, and
    a list build is one workgroup per shape (48 us at 32 shapes) that the forward chain otherwise waits for.
    Returns a list of dicts {fps_idx, new_xyz, sa_idx, blk_idx[, sa_inv, blk_inv]} and the event-free join handle (call
    ``join()`` on the consumer stream before the first use)."""
    if _GEOMETRY_ABLATE:
        key = ("pyramid", tuple(xyz.shape), tuple(npoints), tuple(ks), None if dims is None else tuple(dims))
        if key in _ablate_cache:
            return _ablate_cache[key], (lambda: None)
    main = torch.cuda.current_stream(xyz.device)
    side = _side_stream(xyz.device) if overlap else main
    if overlap:
        side.wait_stream(main)
    levels = []
    with torch.cuda.stream(side):
        cur = xyz.detach().contiguous()
        for n_new, (k_sa, k_blk) in zip(npoints, ks):
            fps_idx = pu.furthest_point_sample(cur, n_new)
            new_xyz = pu.gather_rows(cur, fps_idx)
            sa_idx = pu.knn(new_xyz, cur, k_sa)
            blk_idx = pu.knn(new_xyz, new_xyz, k_blk) if k_blk is not None else None
            lv = {"fps_idx": fps_idx, "new_xyz": new_xyz, "sa_idx": sa_idx, "blk_idx": blk_idx}
            if dims is not None and PYRAMID_LISTS:
                d = dims[len(levels)]
                if hip_attention._use_inverse(torch.float32, False, n_new, cur.shape[1], d):
                    lv["sa_inv"] = hip_attention.inverse_lists(sa_idx, cur.shape[1])
                if blk_idx is not None and hip_attention._use_inverse(torch.float32, False, n_new, n_new, d):
                    lv["blk_inv"] = hip_attention.inverse_lists(blk_idx, n_new)
            levels.append(lv)
            cur = new_xyz

    if _GEOMETRY_ABLATE and not torch.cuda.is_current_stream_capturing():
        _ablate_cache[key] = levels

    def join():
        if overlap:
            torch.cuda.current_stream(xyz.device).wait_stream(side)
            for lv in levels:
                for t in lv.values():
                    for u in (t if isinstance(t, tuple) else (t,)):
                        if u is not None:
                            u.record_stream(torch.cuda.current_stream(xyz.device))

    return levels, join


class _GatherRows(torch.autograd.Function):
    """index_points (model/utils.py:58-70) for idx [B,S] or [B,S,K]; backward = scatter-add."""

    @staticmethod
    def forward(ctx, points, idx):
        B, N, C = points.shape
        flat = idx.reshape(B, -1)
        ctx.save_for_backward(flat)
        ctx.n = N
        points = points.contiguous()
        if points.dtype is torch.bfloat16 and C % 2 == 0:
            # a row gather moves bytes: bf16 rows of C channels are fp32 rows of C / 2 words
            out = pu.gather_rows(points.view(torch.float32), flat).view(torch.bfloat16)
        else:
            out = pu.gather_rows(points.float(), flat).to(points.dtype)
        return out.reshape(*idx.shape, C)

    @staticmethod
    def backward(ctx, grad_out):
        (flat,) = ctx.saved_tensors
        B = flat.shape[0]
        g = grad_out.reshape(B, flat.shape[1], -1).contiguous()
        if g.dtype is torch.bfloat16:      # the scatter accumulates in fp32 (a point is gathered up to k times)
            return pu.scatter_add_rows(g.float(), flat, ctx.n).to(torch.bfloat16), None
        return pu.scatter_add_rows(g, flat, ctx.n), None


def index_points(points: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    return _GatherRows.apply(points, idx)


REL4 = os.environ.get("NSDP_REL4", "1") != "0"     # (A/B knob: 0 = gather + subtraction, the K = 3 layer pads its input itself)


def relative_coords(query: torch.Tensor, source: torch.Tensor, idx: torch.Tensor, sign: float = 1.0) -> torch.Tensor:
    """sign * (query_i - source[idx_ij]) for idx [B, n, k]: the input of a position-encoding MLP.  Where no coordinate needs a
    gradient (always, except in FlowArbitrary's second network) this is ONE launch that writes the K = 4 layer's zero-padded
    16-byte rows [B, n, k, 4] directly (pu.rel_coords4) -- the reference's index_points + subtraction, and the pad in front of
    the layer, are four; `linear` takes the padded rows as they are (K = 4 against a [N, 3] weight)."""
    if (REL4 and query.is_cuda and query.dtype is torch.float32 and source.dtype is torch.float32 and idx.dim() == 3
            and not (torch.is_grad_enabled() and (query.requires_grad or source.requires_grad))):
        return pu.rel_coords4(query.detach().contiguous(), source.detach().contiguous(), idx.contiguous(), sign)
    d = query.unsqueeze(2) - index_points(source, idx)
    return d if sign > 0 else -d


# ---------------------------------------------------------------------------------------------
# dense layers
# ---------------------------------------------------------------------------------------------
def linear(x: torch.Tensor, lin, relu: bool = False, relu_in: bool = False, residual=None, grad_sum=None,
           out_f32: bool = False, premasked: bool = False, mask_dx: bool = False,
           init_gather=None, residual_sign: float = 1.0, skip_src=None, skip_dst=None, tail_src=None,
           tail_dst=None, lay: int = 0) -> torch.Tensor:
    """nn.Linear / 1x1 nn.Conv1d on channels-last rows, on the fp32 matrix cores (hip_linear):
    relu?( relu_in?(x) @ W^T + b (+ residual) ).  ``grad_sum``: hip_linear.InputGradSum shared by layers that read
    the same ``x`` (their input gradients are then summed inside the dX GEMMs)."""
    return hip_linear.linear(x, lin.weight, lin.bias, relu_in=relu_in, relu_out=relu, residual=residual,
                             params=True, grad_sum=grad_sum, out_f32=out_f32, premasked=premasked, mask_dx=mask_dx,
                             init_gather=init_gather, residual_sign=residual_sign,
                             skip_src=skip_src, skip_dst=skip_dst, tail_src=tail_src, tail_dst=tail_dst, lay=lay)


def g16_pair(x: torch.Tensor, lin0, lin1, relu_in0: bool = False) -> bool:
    """May the hidden tensor of lin1(relu(lin0(x))) -- which nothing but these two layers reads -- live in the G16 layout of the
    dense-layer kernels (hip_linear.g16_pair_ok)?  The caller then passes lay=LAY_Y to the first layer and lay=LAY_X to the second."""
    if not (hip_linear.G16 and x.is_cuda and x.dtype is torch.float32) or precision.is_bf16():
        return False
    K = x.shape[-1]
    w0 = lin0.weight.squeeze(-1) if lin0.weight.dim() == 3 else lin0.weight
    w1 = lin1.weight.squeeze(-1) if lin1.weight.dim() == 3 else lin1.weight
    if w0.shape[1] != K or w1.shape[1] != w0.shape[0]:
        return False
    train = torch.is_grad_enabled() and (x.requires_grad or lin0.weight.requires_grad or lin1.weight.requires_grad)
    return hip_linear.g16_pair_ok(x.numel() // K, K, w0.shape[0], w1.shape[0], relu_in0=relu_in0, train=train)


def pos_mlp(x: torch.Tensor, seq: nn.Sequential, init_gather=None):
    """Position-encoding MLP ``seq`` = Sequential(Linear(3, d), ReLU, Linear(d, d)) on coordinates ``x`` that need no gradient,
    without its hidden tensor (hip_linear.pos_mlp); None where that form does not apply (the caller takes the two-layer path)."""
    if PAIR_MASK:
        return None
    return hip_linear.pos_mlp(x, seq[0].weight, seq[0].bias, seq[2].weight, seq[2].bias, init_gather=init_gather)


def k4_tail(x: torch.Tensor, seq: nn.Sequential):
    """hip_linear.K4Tail for a position-encoding MLP ``seq`` = Sequential(Linear(3 or 4, d), ReLU, Linear(d, d)) applied to
    coordinates ``x`` that need no gradient (None otherwise): the second layer's dX GEMM then produces the first layer's
    weight gradient in its epilogue, and the gradient of the hidden tensor is never materialised."""
    if (hip_linear.K4_LINK and torch.is_grad_enabled() and not x.requires_grad and not precision.is_bf16() and not PAIR_MASK
            and x.shape[-1] in (3, 4) and seq[0].weight.requires_grad):
        return hip_linear.K4Tail()
    return None


def input_grad_sum(x: torch.Tensor):
    """An InputGradSum for the dense layers that read ``x`` (None when no gradient will flow: eval / no_grad).  Contract
    (hip_linear.InputGradSum): every layer given it takes part in the backward pass."""
    return hip_linear.InputGradSum() if (torch.is_grad_enabled() and x.requires_grad) else None


# A/B knob, default off: 1 = the backward ReLU mask of a Linear -> ReLU -> Linear pair is applied once, by the second
# layer's dX epilogue.  Measured at B = 32: no gain on forward.yaml (47.6 against 47.8 ms on the same box), 2.5 % SLOWER on
# FlowArbitrary (125.5 against 122.3 ms): the mask read in the bf16x3 epilogue stalls the store phase for longer than the
# masked prologue / masked weight-gradient variants cost.
PAIR_MASK = os.environ.get("NSDP_PAIR_MASK", "0") == "1"


def mlp2(x: torch.Tensor, seq: nn.Sequential, grad_sum=None) -> torch.Tensor:
    """nn.Sequential(Linear, ReLU, Linear) (fc_delta / fc_gamma / fc_middle).
    bf16 storage: the ReLU is the SECOND layer's fused input ReLU (the tensor in between holds the pre-activation): same
    values, but in the backward pass the second layer's dX epilogue applies the ReLU mask once, and neither the first
    layer's dX kernel nor its weight gradient has to stream a mask tensor next to dY (those kernels are pure streams).
    fp32 storage keeps the ReLU in the first layer's epilogue (the bf16x3 GEMM's ReLU prologue costs more than the mask
    operand it saves: +1.9 ms of GEMM time per B = 32 step against -1.2 ms of weight-gradient time).  PAIR_MASK (off by
    default, measured a loss) moves only the BACKWARD mask: the second layer's dX kernel applies (h > 0) in its epilogue
    (`mask_dx`), so the first layer's dX kernel and weight gradient take an already masked gradient (`premasked`)."""
    if precision.is_bf16():
        return linear(linear(x, seq[0], grad_sum=grad_sum), seq[2], relu_in=True)
    if PAIR_MASK and torch.is_grad_enabled():
        return linear(linear(x, seq[0], relu=True, grad_sum=grad_sum, premasked=True), seq[2], mask_dx=True)
    if grad_sum is None and x.shape[-1] in (3, 4):
        y = pos_mlp(x, seq)
        if y is not None:
            return y
    tl = k4_tail(x, seq) if grad_sum is None else None
    g16 = tl is None and g16_pair(x, seq[0], seq[2])      # the hidden tensor and its gradient: G16 layout (private to the pair)
    return linear(linear(x, seq[0], relu=True, grad_sum=grad_sum, tail_src=tl, lay=hip_linear.LAY_Y if g16 else 0), seq[2],
                  tail_dst=tl, lay=hip_linear.LAY_X if g16 else 0)


def batch_norm(x: torch.Tensor, bn: nn.BatchNorm1d, addend=None, relu: bool = False) -> torch.Tensor:
    """relu?( BatchNorm1d(x + addend) ) over the (B*n) rows of a channels-last tensor: batch statistics +
    running update when training, running statistics in eval.  The residual add in front and the ReLU
    behind are fused into the HIP kernels (hip_batchnorm)."""
    if x.dtype is torch.bfloat16 and not hip_batchnorm.NATIVE_BF16:
        a = None if addend is None else addend.float()
        return hip_batchnorm.batch_norm(x.float(), bn, addend=a, relu=relu).to(torch.bfloat16)
    return hip_batchnorm.batch_norm(x, bn, addend=addend, relu=relu)


# ---------------------------------------------------------------------------------------------
# point-transformer vector attention
# ---------------------------------------------------------------------------------------------
SKIP_GRAD = os.environ.get("NSDP_SKIP_GRAD", "1") != "0"       # (A/B knob: 0 = autograd adds the skip connection's gradient)


def skip_grad(x):
    """hip_linear.SkipGrad for a residual block whose input is ``x`` (None when nothing is to be handed over)."""
    if SKIP_GRAD and torch.is_grad_enabled() and x.requires_grad and not precision.is_bf16():
        return hip_linear.SkipGrad()
    return None


FUSE_DPOS = os.environ.get("NSDP_FUSE_DPOS", "1") != "0"     # (A/B knob: 0 = attn_pre_bwd accumulates d(pos) itself)
# u = q_i - k_j + delta(rel_ij) straight out of the position-encoding MLP's last GEMM (its accumulators start from the
# gathered q - k rows: hip_linear's init_gather): no attn_pre pass, and `pos` is never materialised -- the values
# v_j + pos_ij are rebuilt from u inside attn_post (hip_attention._AttnPost, sub=).  fp32 storage, layers on the bf16x3
# kernel.  NSDP_FUSE_PRE=0: the separate attn_pre pass (A/B knob).
FUSE_PRE = os.environ.get("NSDP_FUSE_PRE", "1") != "0"


class PosAsU:
    """What vector_attention hands back as `pos` when it never materialised it: y = q1_i - k1_j + pos (the values the first
    attention's GEMM produced; autograd-wise y IS pos) and the constants (q1, k1) that turn it back into pos."""
    __slots__ = ("y", "q", "kf")

    def __init__(self, y, q, kf):
        self.y, self.q, self.kf = y, q, kf


COMBINE_TABLES = os.environ.get("NSDP_COMBINE_TABLES", "1") != "0"     # (A/B knob: 0 = vector_attention forms v + k etc. itself)


def fused_pre_applies(idx, d) -> bool:
    """Will vector_attention (per-point queries, pos=None) take u straight out of the position-encoding GEMM for this index set
    [B, n, k] and width?  Callers that know can hand it COMBINED tables (see there) and save the elementwise passes."""
    B, n, k = idx.shape
    return bool(FUSE_PRE and COMBINE_TABLES and not precision.is_bf16() and not PAIR_MASK and hip_linear.gather_init_ok(B * n * k, d, d))


def vector_attention(rel, q, kf, vf, idx, fc_delta, fc_gamma, residual=None, pos=None, a_g=None, v_g=None, combined=False,
                     inv=None):
    """sum_j softmax_j[gamma(q_i - kf[idx_ij] + delta(rel_ij))] * (vf[idx_ij] + delta(rel_ij)) (+ residual),
    softmax over the neighbour axis independently per channel (vector attention).

    rel [B,n,k,3] relative coordinates; q [B,n,d] (None: pos_only block -- logits = gamma(delta), values =
    delta); kf, vf [B,N,d] projected source features (gathered inside the fused kernels, never materialised
    as [B,n,k,d]); idx [B,n,k] int32; ``pos`` re-uses an already computed delta(rel) (second attention of the
    set abstraction); a_g / v_g [B,d]: logits / values of a per-shape global token (decoder).
    Returns (aggregate [B,n,d], pos [B,n,k,d])."""
    if isinstance(pos, PosAsU):
        # second attention over the same index set (set abstraction): pos = y - q1_i + k1_j, so
        # u2 = q2_i - k2_j + pos = (q2 - q1)_i - (k2 - k1)_j + y, and the values are y + (v2 + k1)_j - q1_i
        # `combined`: the caller's projections already produced q2 - q1, k2 - k1 and v2 + k1 (signed residuals of their GEMMs)
        y, q1, k1 = pos.y, pos.q, pos.kf
        link = hip_attention.pos_grad_link() if y.requires_grad else None
        if link is not None and FUSE_DPOS:
            link.grad_sum = hip_linear.InputGradSum()
        inv = inv if inv is not None else hip_attention.backward_lists(idx, y.shape[1], kf.shape[1], y.shape[-1])
        u = hip_attention.attn_pre(q, kf, y, idx, link, inv) if combined else hip_attention.attn_pre(q - q1, kf - k1, y, idx, link, inv)
        logits = mlp2(u, fc_gamma, grad_sum=link.grad_sum if link is not None else None)
        out = hip_attention.attn_post(logits, vf, y, idx, a_g=a_g, v_g=v_g, residual=residual, link=link, inv=inv,
                                      sub=(None if combined else k1, q1))
        return out, pos
    if (pos is None and q is not None and FUSE_PRE and not precision.is_bf16() and not PAIR_MASK
            and hip_linear.gather_init_ok(rel.shape[0] * rel.shape[1] * rel.shape[2], fc_delta[2].weight.shape[0],
                                          fc_delta[2].weight.shape[1])):
        B, n, k = idx.shape
        d = fc_delta[2].weight.shape[0]
        per_shape = q.shape[1] == 1 and n != 1
        qd, kd = q.detach(), kf.detach()
        if per_shape:      # one query per shape: the kernel gathers rows of the small table q - k (its look-ahead form)
            gather = (None, 1, (qd - kd).reshape(-1, d), idx.reshape(-1), n * k, kf.shape[1])
        else:
            gather = (qd.reshape(-1, d).contiguous(), k, kd.reshape(-1, d).contiguous(), idx.reshape(-1), n * k, kf.shape[1])
        y = pos_mlp(rel, fc_delta, init_gather=gather)                         # values: u; for autograd this node is `pos`
        if y is None:
            tl = k4_tail(rel, fc_delta)
            h = linear(rel, fc_delta[0], relu=True, tail_src=tl)
            y = linear(h, fc_delta[2], init_gather=gather, tail_dst=tl)
        link = hip_attention.pos_grad_link() if y.requires_grad else None
        if link is not None and FUSE_DPOS:
            link.grad_sum = hip_linear.InputGradSum()
        inv = inv if inv is not None else hip_attention.backward_lists(idx, n, kf.shape[1], d, qb=per_shape)
        u = hip_attention.attn_pre(q, kf, y, idx, link, inv, precomputed=y.detach())      # records the backward, launches nothing
        logits = mlp2(u, fc_gamma, grad_sum=link.grad_sum if link is not None else None)
        # (`combined`: vf is already v + k -- the value projection took k as its residual)
        out = hip_attention.attn_post(logits, vf, y, idx, a_g=a_g, v_g=v_g, residual=residual, link=link, inv=inv,
                                      sub=(None if combined else kd, qd))
        return out, PosAsU(y, qd, kd)
    if combined:
        raise ValueError("combined tables were prepared for the fused path (fused_pre_applies), which this call does not take")
    if pos is None:
        pos = mlp2(rel, fc_delta)                              # 2 dense layers on [B*n*k] rows
    if q is None:
        logits = mlp2(pos, fc_gamma)
        out = hip_attention.attn_post(logits, None, pos, idx, residual=residual)
    else:
        link = hip_attention.pos_grad_link() if pos.requires_grad else None
        if link is not None and FUSE_DPOS and hip_attention.native(pos):
            # d(pos) = d(u) + d(pos)|values is formed by the gamma MLP's first dX GEMM (residual operand), see _PosGrad
            link.grad_sum = hip_linear.InputGradSum()
        # (inverse neighbour lists for the scatters of the backward pass: built here, once per index set)
        if inv is None:
            inv = hip_attention.backward_lists(idx, pos.shape[1], kf.shape[1], pos.shape[-1], qb=(q.shape[1] == 1 and pos.shape[1] != 1))
        u = hip_attention.attn_pre(q, kf, pos, idx, link, inv)      # q_i - kf[idx] + pos, gather fused
        logits = mlp2(u, fc_gamma, grad_sum=link.grad_sum if link is not None else None)
        out = hip_attention.attn_post(logits, vf, pos, idx, a_g=a_g, v_g=v_g, residual=residual, link=link, inv=inv)
    return out, pos
