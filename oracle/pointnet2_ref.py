"""ctypes front-end of oracle/pointnet2_ref.c -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product (nsdp_amd) never does.  All functions take/return numpy arrays on the host.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libnsdp_oracle.so")
_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int32)


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "pointnet2_ref.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_f32p)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_i32p)


def opt_n_threads(work_size: int) -> int:
    return int(lib().nsdp_ref_opt_n_threads(int(work_size)))


def furthest_point_sampling(xyz: np.ndarray, npoint: int) -> np.ndarray:
    """(B,N,3) f32 -> (B,npoint) i32; sampling_gpu.cu:69-173."""
    xyz, px = _f(xyz)
    b, n, _ = xyz.shape
    out = np.zeros((b, npoint), dtype=np.int32)
    rc = lib().nsdp_ref_furthest_point_sampling(b, n, int(npoint), px, out.ctypes.data_as(_i32p))
    assert rc == 0, rc
    return out


def gather_points(points, idx):
    points, pp = _f(points)
    idx, pi = _i(idx)
    b, c, n = points.shape
    m = idx.shape[1]
    out = np.zeros((b, c, m), dtype=np.float32)
    lib().nsdp_ref_gather_points(b, c, n, m, pp, pi, out.ctypes.data_as(_f32p))
    return out


def gather_points_grad(grad_out, idx, n):
    grad_out, pg = _f(grad_out)
    idx, pi = _i(idx)
    b, c, m = grad_out.shape
    out = np.zeros((b, c, n), dtype=np.float32)
    lib().nsdp_ref_gather_points_grad(b, c, int(n), m, pg, pi, out.ctypes.data_as(_f32p))
    return out


def group_points(points, idx):
    points, pp = _f(points)
    idx, pi = _i(idx)
    b, c, n = points.shape
    _, npoints, nsample = idx.shape
    out = np.zeros((b, c, npoints, nsample), dtype=np.float32)
    lib().nsdp_ref_group_points(b, c, n, npoints, nsample, pp, pi, out.ctypes.data_as(_f32p))
    return out


def group_points_grad(grad_out, idx, n):
    grad_out, pg = _f(grad_out)
    idx, pi = _i(idx)
    b, c, npoints, nsample = grad_out.shape
    out = np.zeros((b, c, n), dtype=np.float32)
    lib().nsdp_ref_group_points_grad(b, c, int(n), npoints, nsample, pg, pi,
                                     out.ctypes.data_as(_f32p))
    return out


def ball_query(new_xyz, xyz, radius, nsample):
    new_xyz, pn = _f(new_xyz)
    xyz, px = _f(xyz)
    b, n, _ = xyz.shape
    m = new_xyz.shape[1]
    out = np.zeros((b, m, nsample), dtype=np.int32)
    lib().nsdp_ref_ball_query(b, n, m, ctypes.c_float(radius), int(nsample), pn, px,
                              out.ctypes.data_as(_i32p))
    return out


def three_nn(unknown, known):
    unknown, pu = _f(unknown)
    known, pk = _f(known)
    b, n, _ = unknown.shape
    m = known.shape[1]
    dist2 = np.zeros((b, n, 3), dtype=np.float32)
    idx = np.zeros((b, n, 3), dtype=np.int32)
    lib().nsdp_ref_three_nn(b, n, m, pu, pk, dist2.ctypes.data_as(_f32p),
                            idx.ctypes.data_as(_i32p))
    return dist2, idx


def three_interpolate(points, idx, weight):
    points, pp = _f(points)
    idx, pi = _i(idx)
    weight, pw = _f(weight)
    b, c, m = points.shape
    n = idx.shape[1]
    out = np.zeros((b, c, n), dtype=np.float32)
    lib().nsdp_ref_three_interpolate(b, c, m, n, pp, pi, pw, out.ctypes.data_as(_f32p))
    return out


def three_interpolate_grad(grad_out, idx, weight, m):
    grad_out, pg = _f(grad_out)
    idx, pi = _i(idx)
    weight, pw = _f(weight)
    b, c, n = grad_out.shape
    out = np.zeros((b, c, m), dtype=np.float32)
    lib().nsdp_ref_three_interpolate_grad(b, c, n, int(m), pg, pi, pw, out.ctypes.data_as(_f32p))
    return out


def knn(query, source, k, return_dist=False):
    """k nearest `source` points per `query` point, ascending (dist, idx); model/utils.py:39-55 +
    argsort()[:, :, :k]."""
    query, pq = _f(query)
    source, ps = _f(source)
    b, n, _ = query.shape
    m = source.shape[1]
    idx = np.zeros((b, n, k), dtype=np.int32)
    d2 = np.zeros((b, n, k), dtype=np.float32)
    rc = lib().nsdp_ref_knn(b, n, m, int(k), pq, ps, idx.ctypes.data_as(_i32p),
                            d2.ctypes.data_as(_f32p))
    assert rc == 0, rc
    return (idx, d2) if return_dist else idx
