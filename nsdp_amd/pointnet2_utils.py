"""Drop-in for the reference's ``pointnet2_ops_lib/pointnet2_ops/pointnet2_utils.py`` on MI355X.

Same public names, argument order, tensor layouts, dtypes and differentiability as the reference
(pointnet2_utils.py:34-379): ``furthest_point_sample``, ``gather_operation``, ``three_nn``,
``three_interpolate``, ``grouping_operation``, ``ball_query`` and the ``QueryAndGroup`` / ``GroupAll``
modules -- so a call site such as model/encoder/blocks.py:283 works unchanged.  The native side is
libnsdp_hip.so (hand-written HIP for gfx950) behind the C ABI of include/nsdp_hip.h instead of the
pybind11 module ``pointnet2_ops._ext`` (_ext-src/src/bindings.cpp:6-19).

Error behaviour mirrors the reference's CHECK_* macros (_ext-src/include/utils.h:5-25): non-contiguous,
wrong-dtype or CPU tensors raise ``RuntimeError`` (NsdpHipError is a RuntimeError).
"""
from __future__ import annotations

import ctypes

import torch
import torch.nn as nn
from torch.autograd import Function

from . import _lib
from ._lib import check, fptr, iptr, lib, on_device, optptr, stream_ptr

_c_int = ctypes.c_int


# ------------------------------------------------------------------------------------------------
# thin functional layer over the C ABI (one function per entry point of include/nsdp_hip.h)
# ------------------------------------------------------------------------------------------------
def _fps(xyz: torch.Tensor, npoint: int) -> torch.Tensor:
    B, N, C = xyz.shape
    if C != 3:
        raise _lib.NsdpHipError("xyz must be (B, N, 3)")
    out = torch.empty((B, int(npoint)), dtype=torch.int32, device=xyz.device)
    tmp = torch.empty((B, N), dtype=torch.float32, device=xyz.device) if N > 8192 else None
    with on_device(xyz):
        check(lib().nsdp_furthest_point_sampling(fptr(xyz, "xyz"), _c_int(B), _c_int(N), _c_int(int(npoint)),
                                                 optptr(tmp), iptr(out), stream_ptr()),
              "nsdp_furthest_point_sampling")
    return out


def knn(query: torch.Tensor, source: torch.Tensor, k: int, return_dist: bool = False):
    """``square_distance(query, source).argsort()[:, :, :k]`` without the n x m matrix.
    query (B,n,3), source (B,m,3) -> idx (B,n,k) int32 ascending by (distance, index)."""
    B, n, _ = query.shape
    m = source.shape[1]
    idx = torch.empty((B, n, int(k)), dtype=torch.int32, device=query.device)
    d2 = torch.empty((B, n, int(k)), dtype=torch.float32, device=query.device) if return_dist else None
    with on_device(query):
        check(lib().nsdp_knn(fptr(query, "query"), fptr(source, "source"), _c_int(B), _c_int(n), _c_int(m),
                             _c_int(int(k)), iptr(idx), optptr(d2), stream_ptr()), "nsdp_knn")
    return (idx, d2) if return_dist else idx


def gather_rows(points: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """index_points for a 2-D index: points (B,N,C), idx (B,S) int32 -> (B,S,C)."""
    B, N, C = points.shape
    S = idx.shape[1]
    out = torch.empty((B, S, C), dtype=torch.float32, device=points.device)
    with on_device(points):
        check(lib().nsdp_gather_rows(fptr(points, "points"), iptr(idx, "idx"), _c_int(B), _c_int(N),
                                     _c_int(C), _c_int(S), fptr(out), stream_ptr()), "nsdp_gather_rows")
    return out


def rel_coords4(query: torch.Tensor, source: torch.Tensor, idx: torch.Tensor, sign: float = 1.0) -> torch.Tensor:
    """(sign * (query_i - source[idx_ij]), 0) as [B, n, k, 4] (nsdp_rel_coords4): no gradient."""
    B, n, _ = query.shape
    m, k = source.shape[1], idx.shape[2]
    out = torch.empty((B, n, k, 4), dtype=torch.float32, device=query.device)
    with on_device(query):
        check(lib().nsdp_rel_coords4(fptr(query, "query"), fptr(source, "source"), iptr(idx, "idx"), _c_int(B), _c_int(n),
                                     _c_int(m), _c_int(k), ctypes.c_float(sign), fptr(out), stream_ptr()), "nsdp_rel_coords4")
    return out


# index_points' backward: gather-reduce over inverse index lists (csrc/segment.hip: deterministic, every row read once at
# stream rate) instead of global fp32 atomics (1.0-1.25 TB/s) wherever the lists can be built -- they are cached on the index
# tensor, so an index set that is reused (two gathers of one kNN set, several steps over fixed geometry) builds them once.
# NSDP_SCATTER_ROWS=atomic keeps the atomic kernel (A/B knob).
_SCATTER_INVERSE = __import__("os").environ.get("NSDP_SCATTER_ROWS", "inverse") != "atomic"
_SCATTER_INVERSE_MIN_ROWS = 4096        # rows per launch below which one atomic launch beats invert + segment sum
# ... which is taken only with NSDP_SCATTER_DETERMINISTIC=0: by default a scatter goes through the lists at every size, so that a
# train step's result never depends on the order in which fp32 atomics retire (the replay-equals-eager tests hold every model to that)
_SCATTER_DETERMINISTIC = __import__("os").environ.get("NSDP_SCATTER_DETERMINISTIC", "1") != "0"


def scatter_add_rows(grad_out: torch.Tensor, idx: torch.Tensor, N: int) -> torch.Tensor:
    B, S, C = grad_out.shape
    # (average list length S / N <= 64: the list build sorts every list with one thread -- long lists, e.g. 57 344 gathers
    # of 100 anchors, cost more to build than the atomics cost)
    # Rows of any width: a width that is not a multiple of 4 -- the COORDINATE gradients of FlowArbitrary's second network,
    # whose input points are the first network's predictions (reference model/flow_arbitrary.py:19-27) -- is zero-padded to
    # float4 rows for the list kernel; through the atomic kernel those three-float rows made the whole step depend on the order
    # in which atomics retire (two eager runs of one FlowArbitrary step differed in 1589 of 1813 tensors).  _SCATTER_DETERMINISTIC
    # (default on) also takes the lists below the row count where one atomic launch is faster.
    if (_SCATTER_INVERSE and 1 <= C <= 256 and 0 < int(N) <= 8192 and S <= 64 * int(N) and S > 0
            and (B * S >= _SCATTER_INVERSE_MIN_ROWS or _SCATTER_DETERMINISTIC)
            and grad_out.is_cuda and grad_out.dtype is torch.float32 and grad_out.is_contiguous()):
        from . import hip_attention
        if C % 4:
            padded = torch.nn.functional.pad(grad_out, (0, 4 - C % 4))
            return hip_attention.segment_sum(padded, idx, int(N), 1.0)[:, :, :C].contiguous()
        return hip_attention.segment_sum(grad_out, idx, int(N), 1.0)
    out = torch.empty((B, int(N), C), dtype=torch.float32, device=grad_out.device)
    with on_device(grad_out):
        check(lib().nsdp_scatter_add_rows(fptr(grad_out, "grad_out"), iptr(idx, "idx"), _c_int(B), _c_int(int(N)),
                                          _c_int(C), _c_int(S), fptr(out), stream_ptr()),
              "nsdp_scatter_add_rows")
    return out


# ------------------------------------------------------------------------------------------------
# autograd Functions with the reference's names and signatures
# ------------------------------------------------------------------------------------------------
class FurthestPointSampling(Function):
    @staticmethod
    def forward(ctx, xyz, npoint):
        """xyz (B,N,3) float32 -> (B,npoint) int32 (pointnet2_utils.py:34-62)."""
        out = _fps(xyz, npoint)
        ctx.mark_non_differentiable(out)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        return ()


furthest_point_sample = FurthestPointSampling.apply


def _scatter_cm(grad_out3, idx2, N, idx_obj=None):
    """grad (B,C,N) of a channel-major gather: grad_out3 (B,C,E), idx2 (B,E).  Through the inverse index lists (cached on the
    index tensor; csrc/segment.hip builds them) and the atomics-free LDS-staged kernel where a row of E floats fits LDS and
    the lists are short; None otherwise (the caller then uses the LDS-table / atomic entry point)."""
    B, C, E = grad_out3.shape
    L = lib()
    if not (_SCATTER_INVERSE and 0 < N <= 32768 and E <= 64 * N and L.nsdp_scatter_cm_lists_supported(_c_int(B), _c_int(C),
                                                                                                  _c_int(N), _c_int(E))):
        return None
    from . import hip_attention
    # (the cache of the lists lives on the index tensor OBJECT the caller holds across calls, not on a view of it)
    offsets, entries = hip_attention.inverse_lists(idx2 if idx_obj is None else idx_obj, N)
    grad = torch.empty((B, C, N), dtype=torch.float32, device=grad_out3.device)
    with on_device(grad_out3):
        check(L.nsdp_scatter_cm_lists(fptr(grad_out3, "grad_out"), iptr(offsets), iptr(entries), _c_int(B), _c_int(C), _c_int(N),
                                      _c_int(E), fptr(grad), stream_ptr()), "nsdp_scatter_cm_lists")
    return grad


class GatherOperation(Function):
    @staticmethod
    def forward(ctx, features, idx):
        """features (B,C,N), idx (B,npoint) int32 -> (B,C,npoint) (pointnet2_utils.py:68-101)."""
        ctx.save_for_backward(idx, features)
        B, C, N = features.shape
        M = idx.shape[1]
        out = torch.empty((B, C, M), dtype=torch.float32, device=features.device)
        with on_device(features):
            check(lib().nsdp_gather_points(fptr(features, "features"), iptr(idx, "idx"), _c_int(B), _c_int(C),
                                           _c_int(N), _c_int(M), fptr(out), stream_ptr()), "nsdp_gather_points")
        return out

    @staticmethod
    def backward(ctx, grad_out):
        idx, features = ctx.saved_tensors
        B, C, N = features.shape
        M = idx.shape[1]
        grad_out = grad_out.contiguous()
        grad = _scatter_cm(grad_out, idx, N)
        if grad is not None:
            return grad, None
        grad = torch.empty((B, C, N), dtype=torch.float32, device=grad_out.device)
        with on_device(grad_out):
            check(lib().nsdp_gather_points_grad(fptr(grad_out, "grad_out"), iptr(idx), _c_int(B), _c_int(C),
                                                _c_int(N), _c_int(M), fptr(grad), stream_ptr()),
                  "nsdp_gather_points_grad")
        return grad, None


gather_operation = GatherOperation.apply


class ThreeNN(Function):
    @staticmethod
    def forward(ctx, unknown, known):
        """unknown (B,n,3), known (B,m,3) -> dist (B,n,3) (sqrt of d2), idx (B,n,3) int32
        (pointnet2_utils.py:104-136)."""
        B, n, _ = unknown.shape
        m = known.shape[1]
        dist2 = torch.empty((B, n, 3), dtype=torch.float32, device=unknown.device)
        idx = torch.empty((B, n, 3), dtype=torch.int32, device=unknown.device)
        with on_device(unknown):
            check(lib().nsdp_three_nn(fptr(unknown, "unknown"), fptr(known, "known"), _c_int(B), _c_int(n),
                                      _c_int(m), fptr(dist2), iptr(idx), stream_ptr()), "nsdp_three_nn")
        dist = torch.sqrt(dist2)
        ctx.mark_non_differentiable(dist, idx)
        return dist, idx

    @staticmethod
    def backward(ctx, grad_dist, grad_idx):
        return ()


three_nn = ThreeNN.apply


class ThreeInterpolate(Function):
    @staticmethod
    def forward(ctx, features, idx, weight):
        """features (B,c,m), idx (B,n,3) int32, weight (B,n,3) -> (B,c,n) (pointnet2_utils.py:139-191)."""
        ctx.save_for_backward(idx, weight, features)
        B, c, m = features.shape
        n = idx.shape[1]
        out = torch.empty((B, c, n), dtype=torch.float32, device=features.device)
        with on_device(features):
            check(lib().nsdp_three_interpolate(fptr(features, "features"), iptr(idx, "idx"), fptr(weight, "weight"),
                                               _c_int(B), _c_int(c), _c_int(m), _c_int(n), fptr(out),
                                               stream_ptr()), "nsdp_three_interpolate")
        return out

    @staticmethod
    def backward(ctx, grad_out):
        idx, weight, features = ctx.saved_tensors
        B, c, m = features.shape
        n = idx.shape[1]
        grad_out = grad_out.contiguous()
        grad = torch.empty((B, c, m), dtype=torch.float32, device=grad_out.device)
        L = lib()
        if (_SCATTER_INVERSE and 3 * n <= 64 * m
                and L.nsdp_three_interpolate_grad_lists_supported(_c_int(B), _c_int(c), _c_int(n), _c_int(m))):
            # through the inverse lists of the 3-NN index map (cached on the index tensor): no atomics, deterministic
            from . import hip_attention
            offsets, entries = hip_attention.inverse_lists(idx, m)
            with on_device(grad_out):
                check(L.nsdp_three_interpolate_grad_lists(fptr(grad_out, "grad_out"), fptr(weight, "weight"), iptr(offsets),
                                                          iptr(entries), _c_int(B), _c_int(c), _c_int(n), _c_int(m), fptr(grad),
                                                          stream_ptr()), "nsdp_three_interpolate_grad_lists")
            return grad, torch.zeros_like(idx), torch.zeros_like(weight)
        with on_device(grad_out):
            check(lib().nsdp_three_interpolate_grad(fptr(grad_out, "grad_out"), iptr(idx), fptr(weight), _c_int(B),
                                                    _c_int(c), _c_int(n), _c_int(m), fptr(grad), stream_ptr()),
                  "nsdp_three_interpolate_grad")
        return grad, torch.zeros_like(idx), torch.zeros_like(weight)


three_interpolate = ThreeInterpolate.apply


class GroupingOperation(Function):
    @staticmethod
    def forward(ctx, features, idx):
        """features (B,C,N), idx (B,npoint,nsample) int32 -> (B,C,npoint,nsample)
        (pointnet2_utils.py:194-240)."""
        ctx.save_for_backward(idx, features)
        B, C, N = features.shape
        _, NP, NS = idx.shape
        out = torch.empty((B, C, NP, NS), dtype=torch.float32, device=features.device)
        with on_device(features):
            check(lib().nsdp_group_points(fptr(features, "features"), iptr(idx, "idx"), _c_int(B), _c_int(C),
                                          _c_int(N), _c_int(NP), _c_int(NS), fptr(out), stream_ptr()),
                  "nsdp_group_points")
        return out

    @staticmethod
    def backward(ctx, grad_out):
        idx, features = ctx.saved_tensors
        B, C, N = features.shape
        _, NP, NS = idx.shape
        grad_out = grad_out.contiguous()
        grad = _scatter_cm(grad_out.view(B, C, NP * NS), idx.view(B, NP * NS), N, idx_obj=idx)
        if grad is not None:
            return grad, torch.zeros_like(idx)
        grad = torch.empty((B, C, N), dtype=torch.float32, device=grad_out.device)
        with on_device(grad_out):
            check(lib().nsdp_group_points_grad(fptr(grad_out, "grad_out"), iptr(idx), _c_int(B), _c_int(C),
                                               _c_int(N), _c_int(NP), _c_int(NS), fptr(grad), stream_ptr()),
                  "nsdp_group_points_grad")
        return grad, torch.zeros_like(idx)


grouping_operation = GroupingOperation.apply


class BallQuery(Function):
    @staticmethod
    def forward(ctx, radius, nsample, xyz, new_xyz):
        """xyz (B,N,3), new_xyz (B,npoint,3) -> (B,npoint,nsample) int32 (pointnet2_utils.py:243-276)."""
        B, N, _ = xyz.shape
        M = new_xyz.shape[1]
        out = torch.empty((B, M, int(nsample)), dtype=torch.int32, device=xyz.device)
        with on_device(xyz):
            check(lib().nsdp_ball_query(fptr(new_xyz, "new_xyz"), fptr(xyz, "xyz"), _c_int(B), _c_int(N), _c_int(M),
                                        ctypes.c_float(float(radius)), _c_int(int(nsample)), iptr(out),
                                        stream_ptr()), "nsdp_ball_query")
        ctx.mark_non_differentiable(out)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        return ()


ball_query = BallQuery.apply


def _with_xyz(local_xyz, feats, use_xyz):
    """Channel layout shared by both groupers: relative/absolute coordinates first, then the descriptors."""
    if feats is None:
        return local_xyz
    return torch.cat((local_xyz, feats), dim=1) if use_xyz else feats


class QueryAndGroup(nn.Module):
    """Ball query around each centre, then the members' coordinates (relative to the centre) and descriptors as one
    (B, 3 + C, npoint, nsample) tensor (same contract as pointnet2_utils.py:279-335)."""

    def __init__(self, radius, nsample, use_xyz=True):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz

    def forward(self, xyz, new_xyz, features=None):
        if features is None and not self.use_xyz:
            raise AssertionError("QueryAndGroup: without descriptors the coordinates are the only features "
                                 "(use_xyz=False leaves nothing to return)")
        members = ball_query(self.radius, self.nsample, xyz, new_xyz)                  # (B, npoint, nsample) int32
        centres = new_xyz.permute(0, 2, 1)[:, :, :, None]                               # (B, 3, npoint, 1)
        local = grouping_operation(xyz.permute(0, 2, 1).contiguous(), members) - centres
        picked = grouping_operation(features, members) if features is not None else None
        return _with_xyz(local, picked, self.use_xyz)


class GroupAll(nn.Module):
    """One group holding every point: (B, 3 + C, 1, N); ``new_xyz`` is ignored (pointnet2_utils.py:338-379)."""

    def __init__(self, use_xyz=True):
        super().__init__()
        self.use_xyz = use_xyz

    def forward(self, xyz, new_xyz, features=None):
        everything = xyz.permute(0, 2, 1)[:, :, None, :]                                # (B, 3, 1, N), absolute
        return _with_xyz(everything, None if features is None else features[:, :, None, :], self.use_xyz)
