"""Dense layers on the fp32 matrix cores: autograd wrapper over nsdp_linear_f32 / nsdp_linear_wgrad_f32.

``linear(x, weight, bias, relu_in, relu_out, residual)`` computes
``post(pre(x) @ weight.T + bias (+ residual))`` on channels-last rows; the backward pass runs two more
hand-written MFMA kernels (dX = dY' @ W with the ReLU masks fused into the operand load / epilogue, and
the split-row dW = dY'^T X with the bias gradient as a by-product).  No torch/rocBLAS GEMM is involved.
"""
from __future__ import annotations

import ctypes
import os
import weakref

import torch
import torch.nn.functional as F

from . import precision
from ._lib import check, fptr, lib, on_device, optptr, stream_ptr

_ll = ctypes.c_longlong
_ci = ctypes.c_int


def _fwd(x2, w, b, residual, mask, out_mask, relu_in, relu_out):
    M, K = x2.shape
    N = w.shape[0]
    y = torch.empty((M, N), dtype=torch.float32, device=x2.device)
    with on_device(x2):
        check(lib().nsdp_linear_f32(fptr(x2, "x"), fptr(w, "weight"), optptr(b), optptr(residual), optptr(mask),
                                    optptr(out_mask), fptr(y), _ll(M), _ci(N), _ci(K), _ci(int(relu_in)),
                                    _ci(int(relu_out)), stream_ptr()), "nsdp_linear_f32")
    return y


def pack_weight(w, fwd=True, transposed=False):
    """Fragment-major packs of W [N,K] (nsdp_pack_weight_f32): (Wp or None, WpT or None)."""
    N, K = w.shape
    L = lib()
    L.nsdp_packed_weight_floats.restype = ctypes.c_longlong
    n = int(L.nsdp_packed_weight_floats(_ci(N), _ci(K)))
    wp = torch.empty(n, dtype=torch.float32, device=w.device) if fwd else None
    wpt = torch.empty(n, dtype=torch.float32, device=w.device) if transposed else None
    with on_device(w):
        check(L.nsdp_pack_weight_f32(fptr(w, "weight"), _ci(N), _ci(K), optptr(wp), optptr(wpt), stream_ptr()),
              "nsdp_pack_weight_f32")
    return wp, wpt


def _fwd_wp(x2, wp, N, b, residual, mask, out_mask, relu_in, relu_out):
    """_fwd with the weight given as its fragment-major pack (logical [N, K = x2.shape[1]])."""
    M, K = x2.shape
    y = torch.empty((M, N), dtype=torch.float32, device=x2.device)
    with on_device(x2):
        check(lib().nsdp_linear_wp_f32(fptr(x2, "x"), fptr(wp, "packed weight"), optptr(b), optptr(residual),
                                       optptr(mask), optptr(out_mask), fptr(y), _ll(M), _ci(N), _ci(K),
                                       _ci(int(relu_in)), _ci(int(relu_out)), stream_ptr()), "nsdp_linear_wp_f32")
    return y


def pack_weight_x3(w, fwd=True, transposed=False):
    """bf16x3 split packs of W [N,K] (nsdp_pack_weight_bf16x3): (Wp or None, WpT or None), uint8 buffers."""
    N, K = w.shape
    L = lib()
    L.nsdp_packed_weight_bf16x3_bytes.restype = ctypes.c_longlong
    wp = wpt = None
    if fwd:
        wp = torch.empty(int(L.nsdp_packed_weight_bf16x3_bytes(_ci(N), _ci(K), _ci(0))), dtype=torch.uint8, device=w.device)
    if transposed:
        wpt = torch.empty(int(L.nsdp_packed_weight_bf16x3_bytes(_ci(N), _ci(K), _ci(1))), dtype=torch.uint8, device=w.device)
    with on_device(w):
        check(L.nsdp_pack_weight_bf16x3(fptr(w, "weight"), _ci(N), _ci(K), optptr(wp), optptr(wpt), stream_ptr()),
              "nsdp_pack_weight_bf16x3")
    return wp, wpt


def _fwd_x3(x2, wp, N, b, residual, mask, out_mask, relu_in, relu_out, res_sign=1.0, addend=None):
    """_fwd on the bf16 matrix pipe (3-way split, 6 products): weight given as its bf16x3 pack.
    res_sign = -1: the residual is SUBTRACTED (nsdp_linear_bf16x3_signed_f32; no masks).
    addend: added AFTER the output mask (nsdp_linear_bf16x3_addend_f32; needs mask and out_mask, no input ReLU)."""
    M, K = x2.shape
    y = torch.empty((M, N), dtype=torch.float32, device=x2.device)
    if addend is not None:
        if mask is None or out_mask is None or relu_in or res_sign != 1.0:
            raise ValueError("a post-mask addend needs mask and out_mask, no input ReLU, an unsigned residual")
        with on_device(x2):
            check(lib().nsdp_linear_bf16x3_addend_f32(fptr(x2, "x"), ctypes.c_void_p(wp.data_ptr()), optptr(b), optptr(residual),
                                                      fptr(mask, "mask"), fptr(out_mask, "out_mask"), fptr(addend, "addend"),
                                                      fptr(y), _ll(M), _ci(N), _ci(K), _ci(int(relu_out)), stream_ptr()),
                  "nsdp_linear_bf16x3_addend_f32")
        return y
    if res_sign != 1.0:
        if residual is None or mask is not None or out_mask is not None:
            raise ValueError("a signed residual needs a residual and no masks")
        with on_device(x2):
            check(lib().nsdp_linear_bf16x3_signed_f32(fptr(x2, "x"), ctypes.c_void_p(wp.data_ptr()), optptr(b),
                                                      fptr(residual, "residual"), ctypes.c_float(res_sign), fptr(y), _ll(M),
                                                      _ci(N), _ci(K), _ci(int(relu_in)), _ci(int(relu_out)), stream_ptr()),
                  "nsdp_linear_bf16x3_signed_f32")
        return y
    with on_device(x2):
        check(lib().nsdp_linear_bf16x3_f32(fptr(x2, "x"), ctypes.c_void_p(wp.data_ptr()), optptr(b), optptr(residual),
                                           optptr(mask), optptr(out_mask), fptr(y), _ll(M), _ci(N), _ci(K),
                                           _ci(int(relu_in)), _ci(int(relu_out)), stream_ptr()),
              "nsdp_linear_bf16x3_f32")
    return y


def _fwd_x3_gather(x2, wp, N, b, gather, relu_in, relu_out):
    """_fwd_x3 plus gq[r / g_div] - gk[(r / rows_per_shape) * nsrc + gidx[r]], added in the kernel's epilogue
    (nsdp_linear_bf16x3_gather_f32); gather = (gq [., N], g_div, gk [shapes * nsrc, N], gidx [M] int32, rows_per_shape, nsrc)."""
    gq, g_div, gk, gidx, rps, nsrc = gather            # gq None: gk holds the difference itself (one table, added)
    M, K = x2.shape
    if (gq is not None and gq.shape[-1] != N) or gk.shape[-1] != N or gidx.numel() != M:
        raise ValueError("init_gather: table width / index count do not match the layer")
    y = torch.empty((M, N), dtype=torch.float32, device=x2.device)
    with on_device(x2):
        check(lib().nsdp_linear_bf16x3_gather_f32(fptr(x2, "x"), ctypes.c_void_p(wp.data_ptr()), optptr(b), optptr(gq),
                                                  _ci(int(g_div)), fptr(gk, "gk"), ctypes.c_void_p(gidx.data_ptr()),
                                                  _ci(int(rps)), _ci(int(nsrc)), fptr(y), _ll(M), _ci(N), _ci(K),
                                                  _ci(int(relu_in)), _ci(int(relu_out)), stream_ptr()),
              "nsdp_linear_bf16x3_gather_f32")
    return y


# ------------------------------------------------------------------------------------------------
# G16 layout of the tensors between two dense layers
# ------------------------------------------------------------------------------------------------
# The hidden tensor of a Linear -> ReLU -> Linear pair over [rows, d] per-(centre, neighbour) rows (fc_gamma of every attention
# block, the decoder's ResnetBlockFC) and its gradient are touched by dense-layer kernels only: written by one GEMM, read by the
# next, by a weight gradient and as a ReLU mask.  They live in the G16 layout of csrc/gemm_bf16x3_g16.hip
# ([M / 16][C / 4][16 rows][4 floats]: every wave-wide load / tile store of those kernels is one contiguous KiB, outputs skip the
# LDS staging) -- same values, permuted; the torch tensor that carries one keeps its logical shape and MUST NOT be read by
# anything else (ops.mlp2 / ResnetBlockFC keep it private).  NSDP_G16=0: row-major everywhere (A/B knob).
G16 = os.environ.get("NSDP_G16", "1") != "0"
LAY_X, LAY_Y = 1, 2
# The ReLU mask of a G16 hidden tensor as ONE BIT per element (csrc/x3_kernel.h "ReLU bits": written by the epilogue of the GEMM
# that produces the tensor, read by the masked prologue of its dX GEMM and by its weight gradient) instead of the fp32 tensor
# itself: 28 bytes per row of a 200-wide layer where the mask stream was 800 -- a quarter of the bytes of the decoder's masked dX
# GEMM, a third of its masked weight gradient's.  NSDP_G16_BITS=0: the fp32 tensor as the mask (A/B knob).
G16_BITS = os.environ.get("NSDP_G16_BITS", "1") != "0"


def relu_bits(M, C, device):
    """An uninitialised ReLU-bits buffer for a G16 tensor [M, C] (nsdp_relu_bits_bytes)."""
    L = lib()
    L.nsdp_relu_bits_bytes.restype = ctypes.c_size_t
    return torch.empty(int(L.nsdp_relu_bits_bytes(_ll(M), _ci(C))), dtype=torch.uint8, device=device)


def _fwd_x3_g16(x2, wp, N, b, residual, mask, out_mask, relu_in, relu_out, layout, addend=None, mask_bits=None, bits_out=None):
    """_fwd_x3 with X (and mask) or Y in the G16 layout (nsdp_linear_bf16x3_g16_f32).  ``mask_bits``: the mask of a G16 X as ReLU
    bits (instead of ``mask``); ``bits_out``: a relu_bits() buffer the kernel fills with [Y > 0] (G16 Y with an output ReLU)."""
    M, K = x2.shape
    y = torch.empty((M, N), dtype=torch.float32, device=x2.device)
    with on_device(x2):
        check(lib().nsdp_linear_bf16x3_g16_f32(fptr(x2, "x"), ctypes.c_void_p(wp.data_ptr()), optptr(b), optptr(residual),
                                               optptr(mask), optptr(out_mask), optptr(addend), fptr(y), _ll(M), _ci(N), _ci(K),
                                               _ci(int(relu_in)), _ci(int(relu_out)), _ci(int(layout)),
                                               ctypes.c_void_p(mask_bits.data_ptr() if mask_bits is not None else None),
                                               ctypes.c_void_p(bits_out.data_ptr() if bits_out is not None else None), stream_ptr()),
              "nsdp_linear_bf16x3_g16_f32")
    return y


def to_g16(t, back=False):
    """Row-major [M, C] -> G16 (``back``: the other way), out of place -- tests and debugging."""
    t2 = t.reshape(-1, t.shape[-1]).contiguous()
    out = torch.empty_like(t2)
    with on_device(t2):
        check(lib().nsdp_layout_g16_f32(fptr(t2, "src"), fptr(out), _ll(t2.shape[0]), _ci(t2.shape[1]), _ci(0 if back else 1),
                                        stream_ptr()), "nsdp_layout_g16_f32")
    return out.reshape(t.shape)


def _wgrad_g16_fn(layout, bits=None):
    """Weight-gradient routine (the `fn` protocol of wgrad_direct / _wgrad_deferred) with dY + mask (layout 1) or X (layout 2) in
    the G16 layout (nsdp_linear_wgrad_bf16x3_g16_f32): the sums of _wgrad_x3, bit for bit.  ``bits``: dY's mask as ReLU bits
    (the `mask` argument of the routine is then ignored)."""
    def fn(dy2, x2, mask, relu_x, want_db, out=None):
        if bits is not None:
            mask = None
        M, N = dy2.shape
        K = x2.shape[1]
        L = lib()
        L.nsdp_linear_wgrad_bf16x3_workspace_bytes.restype = ctypes.c_size_t
        nbytes = int(L.nsdp_linear_wgrad_bf16x3_workspace_bytes(_ll(M), _ci(N), _ci(K)))
        ws = torch.empty(max(nbytes // 4, 1), dtype=torch.float32, device=dy2.device)
        dw, db, acc = wgrad_out(out, N, K, want_db, dy2.device)
        batch = _cur_reduce if BATCH_REDUCE > 0 else None
        desc = _ReduceDesc() if batch is not None else None
        with on_device(dy2):
            if batch is not None:
                ptrs = {dw.data_ptr()} | ({db.data_ptr()} if db is not None else set())
                if ptrs & batch["targets"]:
                    _flush_reduce(batch)
            check(L.nsdp_linear_wgrad_bf16x3_g16_f32(fptr(dy2, "dy"), fptr(x2, "x"), optptr(mask), _ci(int(relu_x)), fptr(dw),
                                                     optptr(db), _ll(M), _ci(N), _ci(K), _ci(acc), fptr(ws), ctypes.c_size_t(nbytes),
                                                     ctypes.byref(desc) if desc is not None else None, _ci(int(layout)),
                                                     ctypes.c_void_p(bits.data_ptr() if bits is not None else None), stream_ptr()),
                  "nsdp_linear_wgrad_bf16x3_g16_f32")
            if batch is not None:
                batch["descs"].append(desc)
                batch["keep"].append((ws, dw, db, bits))
                batch["targets"] |= ptrs
                if len(batch["descs"]) + len(batch["descs_b16"]) >= BATCH_REDUCE:
                    _flush_reduce(batch)
        return dw, db
    fn.batched_reduce = True
    fn.extra_tensors = (bits,) if bits is not None else ()      # (read on whatever stream the routine runs: _wgrad_deferred records it)
    return fn


def g16_pair_ok(M, K, H, N, relu_in0=False, train=True):
    """Can the hidden tensor [M, H] of Linear(K, H) -> ReLU -> Linear(H, N) (and, when training, its gradient) live in the G16
    layout?  Every kernel that touches it must have the form: both forward GEMMs, both dX GEMMs, both weight gradients."""
    if not G16 or precision.is_bf16() or M % 16:
        return False
    L = lib()
    ok = (_x3_ok(M, H, K) and _x3_ok(M, N, H)
          and L.nsdp_linear_bf16x3_g16_supported(_ll(M), _ci(H), _ci(K), _ci(LAY_Y), _ci(0), _ci(int(relu_in0)))
          and L.nsdp_linear_bf16x3_g16_supported(_ll(M), _ci(N), _ci(H), _ci(LAY_X), _ci(0), _ci(0)))
    if not ok or not train:
        return bool(ok)
    return bool(_x3_ok(M, H, N) and _x3_ok(M, K, H) and M >= _X3_MIN_ROWS_WGRAD
                and L.nsdp_linear_bf16x3_g16_supported(_ll(M), _ci(H), _ci(N), _ci(LAY_Y), _ci(0), _ci(0))       # second layer dX
                and L.nsdp_linear_bf16x3_g16_supported(_ll(M), _ci(K), _ci(H), _ci(LAY_X), _ci(1), _ci(0))       # first layer dX (masked)
                and L.nsdp_linear_wgrad_bf16x3_g16_supported(_ll(M), _ci(N), _ci(H), _ci(2), _ci(0))
                and L.nsdp_linear_wgrad_bf16x3_g16_supported(_ll(M), _ci(H), _ci(K), _ci(1), _ci(1)))


def gather_init_ok(M, N, K):
    """Can a layer of this shape take ``init_gather`` (it needs the bf16x3 kernel)?"""
    return _x3_ok(M, N, K) and M < 2 ** 31 and not precision.is_bf16()


def wgrad_out(out, N, K, want_db, device):
    """(dw, db, accumulate) for a weight-gradient launch: fresh tensors, or -- `out` = (dW [N,K], db [N] or None) of a
    matching shape -- those buffers with the kernel's accumulate flag (the sum lands in them, no add kernel)."""
    if out is not None:
        dw, db = out
        if (tuple(dw.shape) == (N, K) and dw.dtype is torch.float32 and dw.is_contiguous()
                and (not want_db or (db is not None and db.dtype is torch.float32 and db.is_contiguous()))):
            return dw, (db if want_db else None), 1
    dw = torch.empty((N, K), dtype=torch.float32, device=device)
    db = torch.empty((N,), dtype=torch.float32, device=device) if want_db else None
    return dw, db, 0


def _wgrad(dy2, x2, mask, relu_x, want_db, out=None):
    """dW, db.  `out`: accumulate into these buffers instead (returned as they are when the shapes allow it: callers test
    `dw is out[0]`)."""
    M, N = dy2.shape
    K = x2.shape[1]
    L = lib()
    if _USE_X3 and M >= _X3_MIN_ROWS_WGRAD and L.nsdp_linear_wgrad_bf16x3_supported(_ll(M), _ci(N), _ci(K)):
        return _wgrad_x3(dy2, x2, mask, relu_x, want_db, out)
    L.nsdp_linear_wgrad_workspace_bytes.restype = ctypes.c_size_t
    nbytes = int(L.nsdp_linear_wgrad_workspace_bytes(_ll(M), _ci(N), _ci(K)))
    ws = torch.empty(max(nbytes // 4, 1), dtype=torch.float32, device=dy2.device)
    dw, db, acc = wgrad_out(out, N, K, want_db, dy2.device)
    batch = _cur_reduce
    if batch is not None and BATCH_REDUCE > 0 and M > 0:
        # the small layers' exact-fp32 kernels hand their partials to the pass's batch as well (same protocol as _wgrad_x3)
        ptrs = {dw.data_ptr()} | ({db.data_ptr()} if db is not None else set())
        with on_device(dy2):
            if ptrs & batch["targets"]:
                _flush_reduce(batch)
            desc = _ReduceDescB16()
            check(L.nsdp_linear_wgrad_partials_f32(fptr(dy2, "dy"), fptr(x2, "x"), optptr(mask), _ci(int(relu_x)), fptr(dw),
                                                   optptr(db), _ll(M), _ci(N), _ci(K), _ci(acc), fptr(ws),
                                                   ctypes.c_size_t(nbytes), ctypes.byref(desc), stream_ptr()),
                  "nsdp_linear_wgrad_partials_f32")
            if desc.S == 0:          # few rows: the output-stationary kernel wrote dW / db itself, nothing is pending
                return dw, db
            batch["descs_b16"].append(desc)
            batch["keep"].append((ws, dw, db))
            batch["targets"] |= ptrs
            if len(batch["descs"]) + len(batch["descs_b16"]) >= BATCH_REDUCE:
                _flush_reduce(batch)
        return dw, db
    with on_device(dy2):
        check(L.nsdp_linear_wgrad_f32(fptr(dy2, "dy"), fptr(x2, "x"), optptr(mask), _ci(int(relu_x)), fptr(dw),
                                      optptr(db), _ll(M), _ci(N), _ci(K), _ci(acc), fptr(ws),
                                      ctypes.c_size_t(nbytes), stream_ptr()), "nsdp_linear_wgrad_f32")
    return dw, db


class _ReduceDesc(ctypes.Structure):      # NsdpWgradReduceDesc (include/nsdp_hip.h)
    _fields_ = [("ws", ctypes.c_void_p), ("dW", ctypes.c_void_p), ("db", ctypes.c_void_p), ("S", ctypes.c_int),
                ("nta", ctypes.c_int), ("ktb", ctypes.c_int), ("N", ctypes.c_int), ("K", ctypes.c_int),
                ("accumulate", ctypes.c_int), ("reserved", ctypes.c_int)]


# The partial sums of the side stream's bf16x3 weight gradients are reduced in batches: up to this many layers per reduce
# launch (NSDP_WGRAD_BATCH_REDUCE; 0 = one reduce launch behind every layer's row kernel).  The 98 tiny reduce launches of a
# B = 32 step take 8.6 us each alone and 33 us in the step, where they wait for compute units behind the critical chain;
# dropping them altogether (ablation NSDP_WG3_SKIP_REDUCE=1, wrong gradients) is worth 0.65 ms at B = 32 and 0.8 ms at B = 8.
# Not one batch per step: the partials of a batch stay in HBM until its reduce (~20 MB per layer), and the LAST batch is
# reduced at the very tail of the backward pass, where nothing runs beside it.
BATCH_REDUCE = int(os.environ.get("NSDP_WGRAD_BATCH_REDUCE", "16"))
_cur_reduce = None      # the pending batch of the backward pass whose side-stream section is executing (see _wgrad_deferred)


class _ReduceDescB16(ctypes.Structure):      # NsdpWgradB16ReduceDesc (bf16-storage weight gradients, hip_linear_bf16.wgrad)
    _fields_ = [("ws", ctypes.c_void_p), ("dW", ctypes.c_void_p), ("db", ctypes.c_void_p), ("S", ctypes.c_int),
                ("N", ctypes.c_int), ("K", ctypes.c_int), ("accumulate", ctypes.c_int), ("reserved", ctypes.c_int)]


def _new_reduce_batch():
    return {"descs": [], "descs_b16": [], "keep": [], "targets": set()}


def _flush_reduce(batch):
    """Launch the batched reduce of everything pending (current stream = the one the row kernels ran on)."""
    n = len(batch["descs"])
    if n:
        arr = (_ReduceDesc * n)(*batch["descs"])
        check(lib().nsdp_wgrad_bf16x3_reduce_batched(arr, _ci(n), stream_ptr()), "nsdp_wgrad_bf16x3_reduce_batched")
    n = len(batch["descs_b16"])
    if n:
        arr = (_ReduceDescB16 * n)(*batch["descs_b16"])
        check(lib().nsdp_wgrad_bf16_reduce_batched(arr, _ci(n), stream_ptr()), "nsdp_wgrad_bf16_reduce_batched")
    batch["descs"].clear()
    batch["descs_b16"].clear()
    batch["keep"].clear()
    batch["targets"].clear()


def _wgrad_x3(dy2, x2, mask, relu_x, want_db, out=None):
    """_wgrad on the bf16 matrix pipe (3-way split of both operands, nsdp_linear_wgrad_bf16x3_f32)."""
    M, N = dy2.shape
    K = x2.shape[1]
    L = lib()
    L.nsdp_linear_wgrad_bf16x3_workspace_bytes.restype = ctypes.c_size_t
    nbytes = int(L.nsdp_linear_wgrad_bf16x3_workspace_bytes(_ll(M), _ci(N), _ci(K)))
    ws = torch.empty(max(nbytes // 4, 1), dtype=torch.float32, device=dy2.device)
    dw, db, acc = wgrad_out(out, N, K, want_db, dy2.device)
    batch = _cur_reduce
    if batch is not None and BATCH_REDUCE > 0:
        # row kernel now, reduction with the batch: dw / db hold NOTHING until _flush_reduce ran (the end-of-backward callback
        # flushes before it publishes; _wgrad_deferred flushes before it touches a buffer itself)
        ptrs = {dw.data_ptr()} | ({db.data_ptr()} if db is not None else set())
        if ptrs & batch["targets"]:          # a parameter used twice in the graph: its first reduction must land first
            with on_device(dy2):
                _flush_reduce(batch)
        desc = _ReduceDesc()
        with on_device(dy2):
            check(L.nsdp_linear_wgrad_bf16x3_partials_f32(fptr(dy2, "dy"), fptr(x2, "x"), optptr(mask), _ci(int(relu_x)), fptr(dw),
                                                          optptr(db), _ll(M), _ci(N), _ci(K), _ci(acc), fptr(ws),
                                                          ctypes.c_size_t(nbytes), ctypes.byref(desc), stream_ptr()),
                  "nsdp_linear_wgrad_bf16x3_partials_f32")
            batch["descs"].append(desc)
            batch["keep"].append((ws, dw, db))
            batch["targets"] |= ptrs
            if len(batch["descs"]) + len(batch["descs_b16"]) >= BATCH_REDUCE:
                _flush_reduce(batch)
        return dw, db
    with on_device(dy2):
        check(L.nsdp_linear_wgrad_bf16x3_f32(fptr(dy2, "dy"), fptr(x2, "x"), optptr(mask), _ci(int(relu_x)), fptr(dw),
                                             optptr(db), _ll(M), _ci(N), _ci(K), _ci(acc), fptr(ws),
                                             ctypes.c_size_t(nbytes), stream_ptr()), "nsdp_linear_wgrad_bf16x3_f32")
    return dw, db


# ------------------------------------------------------------------------------------------------
# weight gradients on a side stream
# ------------------------------------------------------------------------------------------------
# dW/db of a layer are needed only by the optimizer, dX is on the critical path of the backward chain.
# When the layer's parameters are known (`params=` of linear(), i.e. the nn.Module call sites), the wgrad
# kernels are launched on a second HIP stream and their results are accumulated there into a per-parameter
# pending buffer; autograd gets no gradient for the weights from this Function.  A callback queued on the
# autograd engine runs when the backward pass has finished: it makes the main stream wait for the side
# stream and only then publishes the buffers to `param.grad` (assign, or `+=` into an existing `.grad` such
# as the data-parallel flat-bucket views).  So nothing on the main stream can observe an unfinished weight
# gradient, and `.grad` is complete when `backward()` returns -- while the wgrad GEMMs overlap the
# (often HBM-bound) attention glue and the dX GEMMs of later layers.
# Without `params` (functional use, torch.autograd.grad on raw tensors) everything runs on the main stream.
# NSDP_WGRAD_STREAM: "1" always, "0" never, "auto" (default): per backward pass, decided by the first (= last
# layer's) weight gradient.  EAGER small batches are launch-bound -- the stream switches, events and record_stream calls of
# ~100 layers cost more host time than the overlap wins (B = 8: 27.3 ms with the side stream, 25.1 ms without; B = 32:
# 52.8 vs 56.3 ms); a step that is being CAPTURED for replay always takes the side stream (see _use_side_stream).
_OVERLAP_WGRAD = os.environ.get("NSDP_WGRAD_STREAM", "auto")
_OVERLAP_WGRAD = {"0": False, "1": True}.get(_OVERLAP_WGRAD, "auto")
_OVERLAP_MIN_ROWS = 131072       # rows of dY at the model's output layer (batch x query points)
# Compute units the side stream's bf16x3 weight-gradient kernels leave free (NSDP_WGRAD_RESERVE_CUS).  Measured at B = 32
# (same box per comparison, two boxes): 0 -> 41.80 / 43.35 ms, 32 -> 41.30 / 42.93, 48 -> 42.90, 64 -> 42.86, 96 -> 42.99,
# 128 -> 43.27: the critical chain's small kernels (the encoder's 100-point levels, BatchNorm, reductions) no longer wait
# for a whole persistent kernel to end before they get a compute unit.
# Re-swept at the end of round 5 (profiles/r5_small_wgrad_and_sweeps.txt): B = 32 32 / 48 / 64 / 80 -> 39.74 / 39.78 / 39.93 / 40.3 ms;
# B = 8 32 / 48 / 64 / 96 -> 13.78 / 13.75 / 13.62 / 13.76: a small batch's critical chain is made of kernels that fill a fraction of
# the chip, and leaving them a quarter of it is worth 0.12 ms.  Unset, the figure follows the pass's size: 64 below
# _RESERVE_SMALL_ROWS rows of dY at the model's output (batch x query points: 65 536 at B = 8), 48 from there on.
SIDE_RESERVE_CUS = int(os.environ["NSDP_WGRAD_RESERVE_CUS"]) if "NSDP_WGRAD_RESERVE_CUS" in os.environ else None
_RESERVE_SMALL_ROWS = 131072
_reserve_now = {}                # (device index, graph task) -> compute units this pass's side-stream launches leave free
_overlap_now = {}                # (device index, graph task) -> decision for that backward pass
_side = {}
_pending = {}          # (device index, graph task) -> {id(param): [param, grad tensor living on the side stream]}
_reduce_batches = {}   # (device index, graph task) -> the pass's pending batch of weight-gradient reductions (_new_reduce_batch)
# Per-backward state is keyed by the autograd graph task that created it: a pass that died with an exception never runs
# its end-of-backward callback, and its leftovers must not be published by (or suppress the callback of) the next pass.


# Two private PyTorch entry points carry the side-stream protocol: the id of the running autograd graph task (keys the
# per-backward state) and the engine's end-of-backward callback queue (publishes the gradients).  If a PyTorch build lacks
# either, weight gradients simply stay on the main stream and are published inside backward -- slower, never wrong.
_graph_task_id = getattr(torch._C, "_current_graph_task_id", None)
_engine = getattr(torch.autograd.Variable, "_execution_engine", None)
_HAVE_ENGINE_HOOKS = _graph_task_id is not None and hasattr(_engine, "queue_callback")


def _pass_key(device):
    return (device.index, _graph_task_id())


def _side_stream(device):
    key = device.index
    if key not in _side:
        _side[key] = torch.cuda.Stream(device=device)
    return _side[key]


def _publish(device, key):
    """End-of-backward callback: join the streams, then hand the pending gradients to the parameters."""
    _overlap_now.pop(key, None)
    _reserve_now.pop(key, None)
    todo = _pending.pop(key, {})
    batch = _reduce_batches.pop(key, None)
    if batch is not None and (batch["descs"] or batch["descs_b16"]):
        with torch.cuda.device(device), torch.cuda.stream(_side_stream(device)):
            _flush_reduce(batch)
    if not todo:
        return
    main = torch.cuda.current_stream(device)
    main.wait_stream(_side_stream(device))
    with torch.no_grad():
        for param, g in todo.values():
            if g is param.grad:          # the kernels accumulated straight into the existing .grad
                continue
            g.record_stream(main)
            if param.grad is None:
                param.grad = g
            else:
                param.grad.add_(g)


def _wgrad_sliced(dy2, x2, mask, relu_x, want_db, k_orig, out=None):
    global _cur_reduce
    padded = x2.shape[1] != k_orig
    prev = _cur_reduce
    if padded:
        _cur_reduce = None             # (the slice below reads dw at once: no pending reduction for it)
    try:
        dw, db = _wgrad(dy2, x2, mask, relu_x, want_db, None if padded else out)
    finally:
        _cur_reduce = prev
    if dw.shape[1] != k_orig:          # zero-padded reduction dimension (K = 3)
        dw = dw[:, :k_orig].contiguous()
    return dw, db


REMASK_K4 = os.environ.get("NSDP_REMASK_K4", "1") != "0"      # (A/B knob: 0 = the K = 4 weight gradient reads the fp32 mask)


def _padded_w4(w_param, batch=True):
    """Row-major [N, 4] copy of a K = 3 / 4 weight (zero-padded), cached on the parameter like the packs.  ``batch=False`` (the
    callers inside a backward pass): a stale copy is rebuilt alone -- the batched rebuild rewrites EVERY pack of the device in place,
    among them the W^T packs this very backward pass saved (a two-graph step advances the weights epoch between its halves)."""
    key = _pack_key(w_param)
    hit = w_param.__dict__.get("_nsdp_w4")
    if hit is not None and hit[0] == key:
        return hit[1]
    w = w_param.detach()
    w = w.squeeze(-1) if w.dim() == 3 else w
    if w.shape[1] < 4 and w.is_contiguous() and w.is_cuda:
        # a stale copy of a registered parameter: rebuilt IN PLACE with every other pack of the device by the one batched launch
        # (five K = 3 layers were ten fill / copy launches per step at the head of the forward chain)
        reg = _pack_registry.get(w.device.index)
        ent = reg["entries"].get((id(w_param), "w4")) if reg is not None else None
        if (batch and hit is not None and ent is not None and ent[0]() is w_param and ent[2] is hit[1]
                and len(reg["entries"]) >= _BATCH_MIN and _repack_all(w.device)):
            return w_param.__dict__["_nsdp_w4"][1]
    w4 = (F.pad(w, (0, 4 - w.shape[1])) if w.shape[1] < 4 else w).contiguous()
    w_param.__dict__["_nsdp_w4"] = (key, w4)
    if w.shape[1] < 4 and w.is_contiguous() and w.is_cuda:
        reg = _pack_registry.setdefault(w.device.index, {"entries": {}, "array": None})
        reg["entries"][(id(w_param), "w4")] = [weakref.ref(w_param), "w4", w4, None]
        reg["array"] = None
    return w4


def _wgrad_k4_remask(w4, bias, k_orig):
    """Weight-gradient routine of a K = 4 layer with a fused output ReLU whose mask is recomputed from the 16-byte input
    rows (nsdp_linear_wgrad_k4_remask_f32) instead of read back from the [M, N] output."""
    def fn(dy2, x2, mask, relu_x, want_db, out=None):
        M, N = dy2.shape
        L = lib()
        L.nsdp_linear_wgrad_workspace_bytes.restype = ctypes.c_size_t
        nbytes = int(L.nsdp_linear_wgrad_workspace_bytes(_ll(M), _ci(N), _ci(4)))
        ws = torch.empty(max(nbytes // 4, 1), dtype=torch.float32, device=dy2.device)
        dw = torch.empty((N, 4), dtype=torch.float32, device=dy2.device)
        db = torch.empty((N,), dtype=torch.float32, device=dy2.device) if want_db else None
        with on_device(dy2):
            check(L.nsdp_linear_wgrad_k4_remask_f32(fptr(dy2, "dy"), fptr(x2, "x"), fptr(w4, "weight"), optptr(bias), fptr(dw),
                                                    optptr(db), _ll(M), _ci(N), _ci(0), fptr(ws), ctypes.c_size_t(nbytes),
                                                    stream_ptr()), "nsdp_linear_wgrad_k4_remask_f32")
        return (dw[:, :k_orig].contiguous() if k_orig < 4 else dw), db
    return fn


def _grad_targets(w_param, b_param, gw, gb):
    """The `out` pair for a weight-gradient launch that should add into the buffers gw / gb (None when one is missing)."""
    if gw is None or (b_param is not None and gb is None) or not gw.is_contiguous():
        return None
    return gw.view(gw.shape[0], -1), gb


def wgrad_direct(dy2, x2, mask, relu_x, k_orig, w_param, b_param, fn=None):
    """Main-stream direct publication: dW / db into `param.grad` (assigned, or accumulated by the kernel itself)."""
    out = _grad_targets(w_param, b_param, w_param.grad, b_param.grad if b_param is not None else None)
    if fn is None:
        gw, gb = _wgrad_sliced(dy2, x2, mask, relu_x, b_param is not None, k_orig, out)
    else:
        gw, gb = fn(dy2, x2, mask, relu_x, b_param is not None, out)
    if out is not None and gw is out[0]:
        return
    with torch.no_grad():
        for prm, g in ((w_param, gw), (b_param, gb)):
            if prm is not None:
                g = g.view_as(prm)
                prm.grad = g if prm.grad is None else prm.grad.add_(g)


# A backward pass split in two autograd passes (nsdp_amd.parallel.GradAllReducer.backward: the decoder's, then the encoder's)
# must decide like ONE pass: the decisions a pass takes at its first weight gradient -- side stream or not, compute units left
# free -- follow the size of the model's OUTPUT layer, which the second pass never sees (its first weight gradient is
# fc_middle's, over one row per shape).  `inherit_pass_decisions()` hands the last pass's decisions to the next one.
_last_decisions = {"overlap": None, "reserve": None}
_inherited = None


class inherit_pass_decisions:
    def __enter__(self):
        global _inherited
        self._was, _inherited = _inherited, dict(_last_decisions)
        return self

    def __exit__(self, *exc):
        global _inherited
        _inherited = self._was
        return False


def _use_side_stream(dy2):
    if not _HAVE_ENGINE_HOOKS:
        return False
    if _OVERLAP_WGRAD != "auto":
        return _OVERLAP_WGRAD
    key = _pass_key(dy2.device)
    use = _overlap_now.get(key)
    if use is None and _inherited is not None and _inherited["overlap"] is not None:
        use = _overlap_now[key] = _last_decisions["overlap"] = _inherited["overlap"]
        if not use:
            dev = dy2.device
            torch.autograd.Variable._execution_engine.queue_callback(lambda: _publish(dev, key))
    if use is None:          # first weight gradient of this backward pass
        # (under stream capture the host-side price of the second stream is paid once, at capture time, while the replay
        # keeps its concurrency -- and at small batches, where kernels do not fill the chip, that concurrency is worth the
        # most: B = 8 replayed 18.3 -> 15.2 ms in fp32, 14.7 -> 12.5 ms with bf16 storage)
        use = _overlap_now[key] = dy2.shape[0] >= _OVERLAP_MIN_ROWS or torch.cuda.is_current_stream_capturing()
        _last_decisions["overlap"] = use
        if not use:          # still need the end-of-backward hook to forget the decision
            dev = dy2.device
            torch.autograd.Variable._execution_engine.queue_callback(lambda: _publish(dev, key))
    return use


def _wgrad_deferred(dy2, x2, mask, relu_x, k_orig, w_param, b_param, fn=None):
    """``fn``: weight-gradient routine (dy2, x2, mask, relu_x, want_db, out) -> (dw, db) of another storage precision.
    The kernels add into what is already there where they can: the pending buffer of a parameter used more than once in the
    graph (FlowArbitrary runs one encoder three times), or an existing `.grad` (the data-parallel flat-bucket views, gradient
    accumulation over micro-batches) -- no add kernels, and nothing to publish for those at the end of the pass."""
    dev = dy2.device
    main = torch.cuda.current_stream(dev)
    side = _side_stream(dev)
    side.wait_stream(main)
    key = _pass_key(dev)
    slot = _pending.get(key)
    if slot is None:
        slot = _pending[key] = {}
        torch.autograd.Variable._execution_engine.queue_callback(lambda: _publish(dev, key))
    with torch.cuda.stream(side):      # everything that touches dw/db before the join stays on `side`
        ent_w = slot.get(id(w_param))
        ent_b = slot.get(id(b_param)) if b_param is not None else None
        if ent_w is None and (b_param is None or ent_b is None):
            # first use in this pass: an existing .grad becomes the pending buffer itself
            gw0 = w_param.grad
            gb0 = b_param.grad if b_param is not None else None
            if gw0 is not None and (b_param is None or gb0 is not None):
                ent_w = slot[id(w_param)] = [w_param, gw0]
                if b_param is not None:
                    ent_b = slot[id(b_param)] = [b_param, gb0]
        out = _grad_targets(w_param, b_param, ent_w[1] if ent_w is not None else None, ent_b[1] if ent_b is not None else None)
        # these launches share the chip with the critical chain on the main stream: the persistent bf16x3 weight-gradient
        # workgroups (one 512-register wave per SIMD) leave SIDE_RESERVE_CUS compute units to it
        L = lib()
        reserve = _reserve_now.get(key)
        if reserve is None and _inherited is not None and _inherited["reserve"] is not None:
            reserve = _reserve_now[key] = _last_decisions["reserve"] = _inherited["reserve"]
        if reserve is None:          # the pass's first weight gradient: the model's output layer
            reserve = _reserve_now[key] = (SIDE_RESERVE_CUS if SIDE_RESERVE_CUS is not None
                                           else (64 if dy2.shape[0] < _RESERVE_SMALL_ROWS else 48))
            _last_decisions["reserve"] = reserve
        L.nsdp_debug_set(_ci(9), _ci(reserve))
        global _cur_reduce
        batch = _reduce_batches.get(key)
        if batch is None:
            batch = _reduce_batches[key] = _new_reduce_batch()
        # (fused routines -- the K = 4 tail, the K = 4 remask kernel -- reduce for themselves; a routine that takes part in the
        # batch says so: hip_linear_bf16's weight gradient)
        _cur_reduce = batch if (fn is None or getattr(fn, "batched_reduce", False)) else None
        if out is not None and batch["targets"] and (
                out[0].data_ptr() in batch["targets"] or (out[1] is not None and out[1].data_ptr() in batch["targets"])):
            # a parameter used twice in the graph (fc_gamma: the neighbours' logits and the global token's) whose first
            # gradient is still a pending reduction: it must land before anything accumulates into the same buffer
            _flush_reduce(batch)
        try:
            if fn is None:
                dw, db = _wgrad_sliced(dy2, x2, mask, relu_x, b_param is not None, k_orig, out)
            else:
                dw, db = fn(dy2, x2, mask, relu_x, b_param is not None, out)
        finally:
            _cur_reduce = None
            L.nsdp_debug_set(_ci(9), _ci(0))
        if out is None or dw is not out[0]:
            for param, g in ((w_param, dw), (b_param, db)):
                if param is None:
                    continue
                g = g.view_as(param)
                ent = slot.get(id(param))
                if ent is None:
                    slot[id(param)] = [param, g]
                else:
                    _flush_reduce(batch)      # (g may still be a pending reduction)
                    ent[1].add_(g)     # (shapes the kernels cannot accumulate in place: the padded K = 3 layers)
    # (everything the side stream's launches read must outlive them for the caching allocator: the operands, and what a routine
    # carries in its closure -- the ReLU bits of a G16 layer; a freed bit buffer reused by the main stream gave a 16 % wrong
    # gradient under NSDP_WGRAD_STREAM=1)
    for t in (dy2, x2, mask) + tuple(getattr(fn, "extra_tensors", ())):
        if t is not None:
            t.record_stream(side)


def _pad_cols(t, mult=4):
    r = (-t.shape[-1]) % mult
    return t if r == 0 else F.pad(t, (0, r))


# Large layers run on the bf16 matrix pipe with the error-compensated 3-way split (nsdp_linear_bf16x3_f32, fp32
# rounding-level accuracy, see csrc/gemm_bf16x3.hip); NSDP_BF16X3=0 keeps every layer on the exact-fp32 MFMA path.
_USE_X3 = os.environ.get("NSDP_BF16X3", "1") != "0"
_X3_MIN_ROWS = int(os.environ.get("NSDP_X3_MIN_ROWS", "32768"))      # (A/B knob)
_X3_MIN_ROWS_WGRAD = 2048      # the split-row wgrad kernel already wins at a few thousand rows


def _x3_ok(M, N, K):
    """Shape contract of the bf16x3 kernel: two k blocks at least, float4 rows, enough rows to fill the chip."""
    return _USE_X3 and M >= _X3_MIN_ROWS and K > 32 and K % 4 == 0 and N % 4 == 0 and N <= 256


# Pack caches are keyed by (storage pointer, tensor version, weights epoch).  The version counter alone is not enough:
# PyTorch's fused optimizers (torch._fused_adam_ ...) and writes through `.data` update parameters WITHOUT bumping it.
# Every optimizer step therefore advances a global epoch (hook on all torch optimizers, registered below); code that
# rewrites weights by other untracked means calls invalidate_weight_packs() itself.
_weights_epoch = 0


def invalidate_weight_packs():
    """Declare every cached weight pack stale (they are rebuilt at their next use)."""
    global _weights_epoch
    _weights_epoch += 1


def _pack_key(t):
    return (t.data_ptr(), t._version, _weights_epoch)


def _param_key(t):
    """Identity of ONE parameter's values between a forward pass and its backward pass (the position-encoding MLPs and the
    K = 4 layers recompute from the parameters in backward what forward computed from them): storage pointer, version counter
    and the number of optimizer steps that touched THIS parameter (`_after_optimizer_step`).  Not the global weights epoch of
    the pack caches: that one advances with every optimizer step of ANY model of the process and with every replay of a
    captured step -- a two-model loop that stepped model B between model A's forward and backward would be told that A's
    first layer had changed."""
    return (t.data_ptr(), t._version, t.__dict__.get("_nsdp_steps", 0))


def _after_optimizer_step(opt):
    invalidate_weight_packs()
    for group in opt.param_groups:
        for prm in group["params"]:
            prm.__dict__["_nsdp_steps"] = prm.__dict__.get("_nsdp_steps", 0) + 1


try:
    from torch.optim.optimizer import register_optimizer_step_post_hook as _reg_post_hook
    _reg_post_hook(lambda _opt, _args, _kwargs: _after_optimizer_step(_opt))
except ImportError:      # (PyTorch without global optimizer hooks: the version counter is all there is)
    pass


class _PackDesc(ctypes.Structure):      # NsdpPackDesc (include/nsdp_hip.h)
    _fields_ = [("W", ctypes.c_void_p), ("Wp", ctypes.c_void_p), ("WpT", ctypes.c_void_p),
                ("N", ctypes.c_int), ("K", ctypes.c_int), ("kind", ctypes.c_int), ("reserved", ctypes.c_int)]


# Every pack that belongs to a parameter is also listed here, so that after an optimizer step ALL of them are rebuilt
# by one nsdp_pack_weights_batched call (in place, same buffers) instead of one launch per layer at its next use.
# device index -> {"entries": {(id(param), kind): [weakref(param), kind, wp, wpt]}, "array": ctypes array or None}
_pack_registry = {}
_BATCH_MIN = 8          # below this many registered packs the per-layer launches are just as good


def _repack_all(device):
    """Rebuild every registered pack of `device` from the current parameter values (current stream), and mark the
    per-parameter caches as valid for the parameters' current (storage pointer, version)."""
    reg = _pack_registry[device.index]
    ents = reg["entries"]
    dead = [k for k, e in ents.items() if e[0]() is None]
    for k in dead:
        del ents[k]
        reg["array"] = None
    live = sorted(ents.values(), key=lambda e: e[1] == "b16")      # fp32 / bf16x3 packs first, then the bf16 ones
    n_b16 = sum(1 for e in live if e[1] == "b16")
    if reg["array"] is None or len(reg["array"]) != len(live):
        reg["array"] = (_PackDesc * len(live))()
    arr = reg["array"]
    for i, (ref, kind, wp, wpt) in enumerate(live):
        prm = ref()
        w = prm.detach()
        w = w.squeeze(-1) if w.dim() == 3 else w
        if not w.is_contiguous():          # (never the case for nn.Linear / 1x1 Conv1d weights)
            return False
        d = arr[i]
        d.W, d.Wp, d.WpT = w.data_ptr(), (wp.data_ptr() if wp is not None else None), (wpt.data_ptr() if wpt is not None else None)
        d.N, d.K, d.kind = w.shape[0], w.shape[1], {"wp": 0, "x3": 1, "b16": 2, "w4": 3}[kind]
    with torch.cuda.device(device):
        n_f = len(live) - n_b16
        if n_f:
            check(lib().nsdp_pack_weights_batched(arr, _ci(n_f), stream_ptr()), "nsdp_pack_weights_batched")
        if n_b16:
            tail = ctypes.cast(ctypes.byref(arr, n_f * ctypes.sizeof(_PackDesc)), ctypes.POINTER(_PackDesc))
            check(lib().nsdp_pack_weights_bf16(tail, _ci(n_b16), stream_ptr()), "nsdp_pack_weights_bf16")
    # the buffers were rewritten in place behind autograd's back: bump their version counters, so that a graph retained
    # from before the optimizer step (it saved a W^T pack for its dX) fails loudly instead of using the new weights
    torch.autograd.graph.increment_version([t for e in live for t in e[2:4] if t is not None])
    for ref, kind, wp, wpt in live:
        prm = ref()
        if kind == "w4":         # (_padded_w4's own cache slot)
            prm.__dict__["_nsdp_w4"] = (_pack_key(prm), wp)
            continue
        cache = prm.__dict__.get("_nsdp_pack")
        key = _pack_key(prm)
        if cache is None or cache["key"] != key:
            cache = prm.__dict__["_nsdp_pack"] = {"key": key}
        cache[kind] = (wp, wpt)
    return True


def registered_packs(device):
    """Strong references to everything a batched rebuild on ``device`` touches right now: (parameter, pack, W^T pack) of every
    registered layer of every LIVE model of the process.  A captured step contains that rebuild as a node with these addresses
    baked in -- whoever keeps the graph must keep them (GraphedStep does): a model of the same process that dies later (another
    test's, a discarded candidate's) would otherwise leave the replays writing packs into freed memory."""
    reg = _pack_registry.get(device.index if device.index is not None else torch.cuda.current_device())
    if reg is None:
        return []
    return [(ent[0](), ent[2], ent[3]) for ent in reg["entries"].values() if ent[0]() is not None]


def refresh_weight_packs(device):
    """Rebuild the registered packs of ``device`` NOW, on the current stream, if any of them is stale.  For code that is
    about to launch dense layers on ANOTHER stream: the batched rebuild is triggered by the first stale pack that is used, on
    whatever stream that use happens to be -- every other stream's layers would then read packs that are being rewritten."""
    reg = _pack_registry.get(device.index)
    if reg is None or len(reg["entries"]) < _BATCH_MIN:
        return
    for ent in reg["entries"].values():
        prm = ent[0]()
        if prm is None:
            continue
        cache = prm.__dict__.get("_nsdp_pack")
        if cache is None or cache["key"] != _pack_key(prm):
            _repack_all(device)
            return


def _packs(w, owner, kind, want_t):
    """(pack of W, pack of W^T or None) of w [N,K]; kind 'wp' = fp32 fragment-major, 'x3' = bf16x3 planes.
    `owner` (the layer's nn.Parameter, or None) carries a cache keyed by (storage pointer, version counter, weights
    epoch): load_state_dict bumps the version, every optimizer step the epoch, .to(device) changes the pointer.  Without an owner
    the pack is rebuilt per call (a reused address of a freed temporary must never hit a stale pack).
    A stale cache of a registered parameter triggers one batched rebuild of every registered pack (_repack_all)."""
    cache = None
    if owner is not None:
        key = _pack_key(w)
        cache = owner.__dict__.get("_nsdp_pack")
        stale = cache is not None and cache["key"] != key
        if cache is None or stale:
            reg = _pack_registry.get(w.device.index)
            ent = reg["entries"].get((id(owner), kind)) if reg is not None else None
            if (stale and ent is not None and ent[0]() is owner and len(reg["entries"]) >= _BATCH_MIN
                    and (ent[3] is not None or not want_t) and _repack_all(w.device)):
                cache = owner.__dict__["_nsdp_pack"]
            else:
                cache = owner.__dict__["_nsdp_pack"] = {"key": key}
        ent = cache.get(kind)
        if ent is not None and (ent[1] is not None or not want_t):
            return ent
    wc = w if w.is_contiguous() else w.contiguous()
    if kind == "b16":
        from .hip_linear_bf16 import pack_weight_b16
        ent = pack_weight_b16(wc, True, want_t)
    else:
        ent = pack_weight_x3(wc, True, want_t) if kind == "x3" else pack_weight(wc, True, want_t)
    if cache is not None:
        cache[kind] = ent
        if wc is w:
            reg = _pack_registry.setdefault(w.device.index, {"entries": {}, "array": None})
            reg["entries"][(id(owner), kind)] = [weakref.ref(owner), kind, ent[0], ent[1]]
            reg["array"] = None
    return ent


def _run(kind, x2, pack, N, b, residual, mask, out_mask, relu_in, relu_out, res_sign=1.0, addend=None, lay=0, mask_bits=None,
         bits_out=None):
    if lay:        # G16 operands (see _fwd_x3_g16): the caller checked the form (g16_pair_ok)
        if kind != "x3" or res_sign != 1.0:
            raise ValueError("G16 layouts belong to the bf16x3 kernels")
        return _fwd_x3_g16(x2, pack, N, b, residual, None if mask_bits is not None else mask, out_mask, relu_in, relu_out, lay,
                           addend=addend, mask_bits=mask_bits, bits_out=bits_out)
    if addend is not None:
        if kind == "x3" and mask is not None and out_mask is not None and not relu_in:
            return _fwd_x3(x2, pack, N, b, residual, mask, out_mask, relu_in, relu_out, addend=addend)
        return _run(kind, x2, pack, N, b, residual, mask, out_mask, relu_in, relu_out, res_sign).add_(addend)
    if res_sign != 1.0:
        if kind == "x3":
            return _fwd_x3(x2, pack, N, b, residual, mask, out_mask, relu_in, relu_out, res_sign)
        residual = -residual          # (the small-M exact-fp32 kernel adds: one elementwise launch on a small tensor)
    fn = _fwd_x3 if kind == "x3" else _fwd_wp
    return fn(x2, pack, N, b, residual, mask, out_mask, relu_in, relu_out)


class InputGradSum:
    """Shared by the dense layers that read the SAME input tensor (the decoder feeds its latent code to six of them).
    Autograd would produce six dX tensors and add them pairwise; with this hand-over each layer's dX GEMM takes the
    running sum as its fused `residual` operand, all but the last report no gradient, and the last one reports the
    total.  Contract: every layer given the link must take part in the backward pass (true when each of their outputs
    feeds the loss, as in the decoder trunk); layers with a fused input ReLU cannot join (their dX is masked after
    the residual is added)."""
    __slots__ = ("total", "pending", "buf")

    def __init__(self):
        self.total = 0
        self.pending = 0
        self.buf = None


class SkipGrad:
    """Hand-over of the skip connection's gradient inside a pre-activation residual block y = x + fc_1(relu(fc_0(relu(x)))):
    fc_1 (``skip_src``: the layer whose `residual` operand is x) deposits d(residual) = dy here instead of reporting it, fc_0
    (``skip_dst``: the layer whose input is x) adds it to its dX behind the input ReLU's mask -- in the dX GEMM's epilogue
    (nsdp_linear_bf16x3_addend_f32) where that kernel runs, with one add otherwise -- and reports the block's whole input
    gradient.  Autograd would add the two [rows, d] tensors with an elementwise kernel per block.  fc_0's forward arms the
    link (only if its input needs a gradient); fc_1's backward always runs before fc_0's (it consumes fc_0's output)."""
    __slots__ = ("armed", "buf")

    def __init__(self):
        self.armed = False
        self.buf = None


# A/B knobs of K4Tail.  NSDP_K4_LINK=0: no link at all (the hidden tensor's gradient goes through autograd: dX GEMM on the main
# stream, the K = 4 weight gradient on the side stream).  NSDP_K4_TAIL=0: linked, but always the two launches (no fused epilogue).
# NSDP_K4_TAIL_SIDE=0: the linked launches stay on the main stream.  Measured at B = 32, one box, three interleaved rounds:
# no link 41.55 / 41.34 / 41.42 ms, fused tail on the main stream 41.54 / 41.79 / 41.71, on the side stream 41.03 / 40.98 / 41.14.
K4_LINK = os.environ.get("NSDP_K4_LINK", "1") != "0"
K4_TAIL = os.environ.get("NSDP_K4_TAIL", "1") != "0"
K4_TAIL_SIDE = os.environ.get("NSDP_K4_TAIL_SIDE", "1") != "0"
# compute units the tail GEMM leaves free on the side stream (None: SIDE_RESERVE_CUS, like the weight-gradient kernels)
K4_TAIL_RESERVE = int(os.environ["NSDP_K4_TAIL_RESERVE"]) if "NSDP_K4_TAIL_RESERVE" in os.environ else None


class K4Tail:
    """Link between the two layers of a position-encoding MLP Linear(3, d) -> ReLU -> Linear(d, d) whose input (relative
    coordinates) needs no gradient.  The gradient of the hidden tensor h0 then has ONE reader, the first layer's weight gradient:
    (a) that dX GEMM is weight-gradient work, not part of the critical chain: it runs on the weight-gradient side stream; (b) where
    the kernel has the form (13- and 16-tile streaming forms) the GEMM forms dW0 / db0 in its epilogue
    (nsdp_linear_bf16x3_k4tail_f32) -- the [rows, d] gradient is neither written nor read back, and the ReLU mask is recomputed
    from the 16-byte coordinate rows.  The first layer (``tail_src``) arms the link and leaves its operands here; the second
    (``tail_dst``, the ONLY reader of h0) takes it, publishes the first layer's weight gradient from its own backward and
    reports no input gradient: the first layer's backward then receives None and does nothing."""
    __slots__ = ("armed", "taken", "x4", "w_param", "b_param", "k_orig", "fwd_key")

    def __init__(self):
        self.armed = self.taken = False
        self.x4 = self.w_param = self.b_param = self.k_orig = self.fwd_key = None


def _k4tail_fn(link, wpt, n_hidden, kind_t, h0):
    """Weight-gradient routine (the `fn` protocol of wgrad_direct / _wgrad_deferred) of the K = 4 layer behind `link`, run from the
    NEXT layer's backward: dy2 is that layer's output gradient [M, K], wpt its W^T pack, h0 its input (the K = 4 layer's output).
    One launch where the dX GEMM has the tail form, otherwise the dX GEMM and the K = 4 weight-gradient kernel back to back --
    on whatever stream the caller put it (nothing on the critical chain reads the hidden tensor's gradient)."""
    def fn(dy2, x4, mask, relu_x, want_db, out=None):
        M, K = dy2.shape
        L = lib()
        b0 = None if link.b_param is None else link.b_param.detach()
        if not (K4_TAIL and kind_t == "x3" and L.nsdp_linear_bf16x3_k4tail_ok(_ll(M), _ci(n_hidden), _ci(K))):
            dh = _run(kind_t, dy2, wpt, n_hidden, None, None, None, None, False, False)
            if REMASK_K4 and n_hidden % 4 == 0 and n_hidden >= 16 and M >= 4096:
                return _wgrad_k4_remask(_padded_w4(link.w_param, batch=False), b0, link.k_orig)(dh, x4, None, False, want_db, out)
            return _wgrad_sliced(dh, x4, h0, False, want_db, link.k_orig, out)
        L.nsdp_linear_bf16x3_k4tail_workspace_bytes.restype = ctypes.c_size_t
        nbytes = int(L.nsdp_linear_bf16x3_k4tail_workspace_bytes(_ll(M), _ci(n_hidden)))
        ws = torch.empty(max(nbytes // 4, 4), dtype=torch.float32, device=dy2.device)
        dw, db, acc = wgrad_out(out, n_hidden, link.k_orig, want_db, dy2.device)
        with on_device(dy2):
            if K4_TAIL_RESERVE is not None and torch.cuda.current_stream(dy2.device) == _side.get(dy2.device.index):
                L.nsdp_debug_set(_ci(9), _ci(K4_TAIL_RESERVE))      # (_wgrad_deferred resets the hint after this routine)
            check(L.nsdp_linear_bf16x3_k4tail_f32(fptr(dy2, "dy"), ctypes.c_void_p(wpt.data_ptr()), fptr(x4, "x4"),
                                                  fptr(_padded_w4(link.w_param, batch=False), "w0"), optptr(b0), fptr(dw), optptr(db), _ll(M),
                                                  _ci(n_hidden), _ci(K), _ci(int(link.k_orig)), _ci(acc), fptr(ws),
                                                  ctypes.c_size_t(nbytes), stream_ptr()), "nsdp_linear_bf16x3_k4tail_f32")
        return dw, db
    return fn


# ------------------------------------------------------------------------------------------------
# Position-encoding MLPs from the coordinates: the hidden tensor is never materialised
# ------------------------------------------------------------------------------------------------
# fc_delta = Linear(3, d) -> ReLU -> Linear(d, d) on [B, n, k, 3] relative coordinates (reference model/encoder/blocks.py:86-90,
# :281-285, model/decoder/blocks.py:30-34).  Its hidden tensor h0 [rows, d] (1.47 GB in the decoder of a B = 32 step) was written
# by the K = 4 kernel, read by the second layer's GEMM, kept for the backward pass and read again by the second layer's weight
# gradient.  With NSDP_H0_RECOMPUTE (default on) the operand producers of those two kernels evaluate the K = 4 layer themselves
# from the 16-byte coordinate rows (nsdp_linear_bf16x3_h0_f32, nsdp_linear_wgrad_bf16x3_h0_f32: the forward kernel's own
# expression, so every value is the one that kernel would have stored); the first layer's weight gradient already came out of the
# second layer's dX GEMM (K4Tail).  What is left of the pair: one GEMM forward, one weight-gradient kernel and the tail GEMM backward.
# Measured (tools/bench_h0.py, profiles/r5_h0_recompute.txt): these GEMMs are not HBM-bound, and the producer's ~16 VALU operations
# per value pair cost about what the activation DMA did -- so the time gained is the K = 4 kernel's minus a slower weight gradient:
# -40 % forward / -13 % forward + backward on the 128-wide blocks, -6 ... -14 % / -2 % on the decoder's 1 M x 200, a LOSS on the
# 256-wide three-row-tile form (+4 % / +2 %) and on 200-wide layers below ~0.5 M rows.  NSDP_H0_RECOMPUTE: 0 off, 1 (default) the
# shape classes where it pays, 2 every shape the kernels support.  What it always saves is the hidden tensor itself (1.47 GB kept
# from forward to backward in the decoder of a B = 32 step).
H0_RECOMPUTE = int(os.environ.get("NSDP_H0_RECOMPUTE", "1"))


def _h0_pays(M, N):
    nt = (N + 15) // 16
    if H0_RECOMPUTE >= 2 or nt <= 8:
        return True
    if nt <= 13:
        return M >= (1 << 19)
    return M <= 65536      # (16 n tiles: the two-row-tile form of the small levels)


def _fwd_x3_h0(x4, w4, b0, wp, N, K, b, gather):
    """relu(x4 w4^T + b0) W^T + b (+ the gathered addend of _fwd_x3_gather) without the hidden tensor (nsdp_linear_bf16x3_h0_f32)."""
    M = x4.shape[0]
    gq = gk = gidx = None
    g_div = rps = nsrc = 0
    if gather is not None:
        gq, g_div, gk, gidx, rps, nsrc = gather
        if (gq is not None and gq.shape[-1] != N) or gk.shape[-1] != N or gidx.numel() != M:
            raise ValueError("init_gather: table width / index count do not match the layer")
    y = torch.empty((M, N), dtype=torch.float32, device=x4.device)
    with on_device(x4):
        check(lib().nsdp_linear_bf16x3_h0_f32(fptr(x4, "x4"), fptr(w4, "w0"), optptr(b0), ctypes.c_void_p(wp.data_ptr()), optptr(b),
                                              optptr(gq), _ci(int(g_div)), optptr(gk),
                                              ctypes.c_void_p(gidx.data_ptr() if gidx is not None else None), _ci(int(rps)),
                                              _ci(int(nsrc)), fptr(y), _ll(M), _ci(N), _ci(K), stream_ptr()),
              "nsdp_linear_bf16x3_h0_f32")
    return y


def _wgrad_h0_fn(w4, b0, K):
    """Weight-gradient routine (the `fn` protocol of wgrad_direct / _wgrad_deferred) of the SECOND layer of a position-encoding
    MLP whose hidden tensor was never stored: x2 is the [M, 4] coordinate rows, (w4, b0) the first layer."""
    def fn(dy2, x4, mask, relu_x, want_db, out=None):
        M, N = dy2.shape
        L = lib()
        L.nsdp_linear_wgrad_bf16x3_workspace_bytes.restype = ctypes.c_size_t
        nbytes = int(L.nsdp_linear_wgrad_bf16x3_workspace_bytes(_ll(M), _ci(N), _ci(K)))
        ws = torch.empty(max(nbytes // 4, 1), dtype=torch.float32, device=dy2.device)
        dw, db, acc = wgrad_out(out, N, K, want_db, dy2.device)
        batch = _cur_reduce if BATCH_REDUCE > 0 else None
        desc = _ReduceDesc() if batch is not None else None
        with on_device(dy2):
            if batch is not None:
                ptrs = {dw.data_ptr()} | ({db.data_ptr()} if db is not None else set())
                if ptrs & batch["targets"]:
                    _flush_reduce(batch)
            check(L.nsdp_linear_wgrad_bf16x3_h0_f32(fptr(dy2, "dy"), fptr(x4, "x4"), fptr(w4, "w0"), optptr(b0), fptr(dw), optptr(db),
                                                    _ll(M), _ci(N), _ci(K), _ci(acc), fptr(ws), ctypes.c_size_t(nbytes),
                                                    ctypes.byref(desc) if desc is not None else None, stream_ptr()),
                  "nsdp_linear_wgrad_bf16x3_h0_f32")
            if batch is not None:
                batch["descs"].append(desc)
                batch["keep"].append((ws, dw, db, x4, w4, b0))
                batch["targets"] |= ptrs
                if len(batch["descs"]) + len(batch["descs_b16"]) >= BATCH_REDUCE:
                    _flush_reduce(batch)
        return dw, db
    fn.batched_reduce = True
    return fn


class _PosMlpFn(torch.autograd.Function):
    """y = relu(x4 W0^T + b0) W1^T + b1 (+ gathered addend) with direct publication of all four parameter gradients; x4 needs no
    gradient.  Saved for the backward pass: the coordinate rows, the padded first layer and the W1^T pack -- not the hidden tensor."""

    @staticmethod
    def forward(ctx, x4, w4, b0, w1, b1, wp, wpt, init_gather, k_orig0, w0_param, b0_param, w1_param, b1_param):
        N, K = w1.shape
        y = _fwd_x3_h0(x4, w4, b0, wp, N, K, b1, init_gather)
        ctx.save_for_backward(x4, w4, b0, wpt)
        ctx.params = (w0_param, b0_param, w1_param, b1_param)
        ctx.k_orig0, ctx.n_out, ctx.hidden = k_orig0, N, K
        ctx.fwd_key = (_param_key(w0_param), None if b0_param is None else _param_key(b0_param))
        return y

    @staticmethod
    def backward(ctx, dy):
        x4, w4, b0, wpt = ctx.saved_tensors
        w0_param, b0_param, w1_param, b1_param = ctx.params
        if ctx.fwd_key != (_param_key(w0_param), None if b0_param is None else _param_key(b0_param)):
            raise RuntimeError("position-encoding MLP: its first layer changed between forward and backward; the hidden tensor "
                               "is recomputed from the parameters (NSDP_H0_RECOMPUTE=0 keeps it)")
        N, K = ctx.n_out, ctx.hidden
        dy2 = dy.reshape(-1, N)
        dy2 = dy2 if dy2.is_contiguous() else dy2.contiguous()
        link = K4Tail()
        link.x4, link.w_param, link.b_param, link.k_orig = x4, w0_param, b0_param, ctx.k_orig0
        fn1 = _wgrad_h0_fn(w4, b0, K)
        tfn = _k4tail_fn(link, wpt, K, "x3", None)
        if _use_side_stream(dy2):
            _wgrad_deferred(dy2, x4, None, False, K, w1_param, b1_param, fn=fn1)
            if K4_TAIL_SIDE:
                _wgrad_deferred(dy2, x4, None, False, ctx.k_orig0, w0_param, b0_param, fn=tfn)
                wpt.record_stream(_side_stream(dy2.device))
            else:
                wgrad_direct(dy2, x4, None, False, ctx.k_orig0, w0_param, b0_param, fn=tfn)
        else:
            wgrad_direct(dy2, x4, None, False, K, w1_param, b1_param, fn=fn1)
            wgrad_direct(dy2, x4, None, False, ctx.k_orig0, w0_param, b0_param, fn=tfn)
        return (None,) * 13


def pos_mlp(x, weight0, bias0, weight1, bias1, init_gather=None):
    """Linear(3 or 4, d) -> ReLU -> Linear(d, N) on coordinates that need no gradient, without the hidden tensor (see above).
    The four arguments are the layers' leaf nn.Parameters (bias0 / bias1 may be None).  Returns None where this form does not
    apply -- the caller then takes the two-layer path: bf16 storage, shapes outside the kernels' range, parameter gradients that
    must go through autograd (hooks, autograd_param_grads), A/B knobs of the K = 4 tail switched off."""
    if not H0_RECOMPUTE or precision.is_bf16() or x.dtype is not torch.float32 or x.shape[-1] not in (3, 4):
        return None
    grad = torch.is_grad_enabled()
    if grad and x.requires_grad:
        return None
    w0 = weight0.squeeze(-1) if weight0.dim() == 3 else weight0
    w1 = weight1.squeeze(-1) if weight1.dim() == 3 else weight1
    N, K = w1.shape
    if w0.shape[0] != K or w0.shape[1] not in (3, 4) or w0.shape[1] > x.shape[-1] or not (weight0.is_leaf and weight1.is_leaf):
        return None
    M = x.numel() // x.shape[-1]
    L = lib()
    if not (_x3_ok(M, N, K) and L.nsdp_linear_bf16x3_h0_supported(_ll(M), _ci(N), _ci(K)) and _h0_pays(M, N)):
        return None
    needs = [p for p in (weight0, bias0, weight1, bias1) if p is not None and p.requires_grad]
    train = grad and bool(needs)
    if train:
        every = [p for p in (weight0, bias0, weight1, bias1) if p is not None]
        if (len(needs) != len(every) or not _PARAM_GRADS_DIRECT or any(_observed(p) or not p.is_leaf for p in every)
                or not (K4_LINK and REMASK_K4) or M < 4096 or not _x3_ok(M, K, N)
                or not L.nsdp_linear_wgrad_bf16x3_h0_supported(_ll(M), _ci(N), _ci(K))):
            return None
    x4 = x.reshape(-1, x.shape[-1])
    x4 = x4 if x4.is_contiguous() else x4.contiguous()
    x4 = _pad_cols(x4)
    w4 = _padded_w4(weight0)
    b0 = None if bias0 is None else bias0.detach()
    b1 = None if bias1 is None else bias1.detach()
    wd = w1.detach()
    wp = _packs(wd, weight1, "x3", train)[0]
    if not train:
        y = _fwd_x3_h0(x4, w4, b0, wp, N, K, b1, init_gather)
    else:
        wpt = _packs(wd, weight1, "x3", True)[1]
        y = _PosMlpFn.apply(x4, w4, b0, wd, b1, wp, wpt, init_gather, int(w0.shape[1]), weight0, bias0, weight1, bias1)
    return y.reshape(*x.shape[:-1], N)


class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, residual, relu_in, relu_out, w_param=None, b_param=None, grad_sum=None, owner=None, bw=0,
                init_gather=None, res_sign=1.0, skip=None, tail=None, lay=0):
        # skip: (SkipGrad, is_src) -- see SkipGrad;  tail: (K4Tail, is_src) -- see K4Tail
        # lay: LAY_Y = this layer's OUTPUT (and the gradient arriving for it) is in the G16 layout, LAY_X = its INPUT (and the
        # gradient it reports) is -- the two layers around a private hidden tensor (g16_pair_ok)
        ctx.lay = lay
        ctx.tail = None
        ctx.skip_src = ctx.skip_dst = None
        if skip is not None:
            link, is_src = skip
            if is_src:
                if residual is None or relu_out:
                    raise ValueError("skip_src: the layer that adds x as its residual, without an output ReLU")
                if link.armed and ctx.needs_input_grad[3]:
                    ctx.skip_src = link
            elif ctx.needs_input_grad[0]:
                if grad_sum is not None:
                    raise ValueError("skip_dst and InputGradSum do not combine")
                link.armed = True
                ctx.skip_dst = link
        # res_sign: -1 = the residual is subtracted (a projection "minus a table" in one launch)
        # init_gather: (gq, g_div, gk, gidx, rows_per_shape, nsrc), constants of this node -- a gathered difference of two small
        # tables joins the output in the epilogue (see _fwd_x3_gather); whoever owns the tables accounts for their gradients
        # bw (backward contract of a Linear -> ReLU -> Linear pair whose middle tensor has no other reader):
        #   1 on the first layer ("premasked"): the incoming gradient already carries this layer's ReLU mask
        #   2 on the second ("mask_dx"): dX is masked by (x > 0) in the dX kernel's epilogue, i.e. it IS that gradient
        # -- the mask is then streamed once (as the out_mask of one kernel) instead of twice (masked prologue of the first
        # layer's dX kernel + mask operand of its weight gradient), and nothing is saved for it.
        ctx.premasked, ctx.mask_dx = bool(bw & 1) and relu_out, bool(bw & 2)
        ctx.w_param, ctx.b_param = w_param, b_param
        owner = w_param if w_param is not None else owner     # whose __dict__ carries the pack cache
        ctx.grad_sum = None
        if grad_sum is not None and ctx.needs_input_grad[0]:
            if relu_in or (bw & 2) or x.shape[-1] % 4:
                raise ValueError("InputGradSum: layers with a fused input ReLU or a padded input width cannot join")
            grad_sum.total += 1
            grad_sum.pending += 1
            ctx.grad_sum = grad_sum
        K = x.shape[-1]
        N = w.shape[0]
        if K != w.shape[1]:
            # relative coordinates that arrive zero-padded to the K = 4 kernels' 16-byte rows (ops.relative_coords) against the
            # layer's [N, 3] weight: nothing to pad here; every gradient shape below follows the WEIGHT
            if not (K == 4 and w.shape[1] == 3) or ctx.needs_input_grad[0]:
                raise ValueError(f"linear: input width {K} against a [{N}, {w.shape[1]}] weight")
        x2 = x.reshape(-1, K)
        x2 = x2 if x2.is_contiguous() else x2.contiguous()
        K = w.shape[1]
        if K % 4 and x2.shape[1] % 4:  # K = 3 (relative coordinates): zero-pad the reduction dimension (the weight pack pads itself)
            x2 = _pad_cols(x2)
        res2 = None
        if residual is not None:
            res2 = residual.reshape(-1, N)
            res2 = res2 if res2.is_contiguous() else res2.contiguous()
        M, Kp = x2.shape
        want_t = bool(ctx.needs_input_grad[0])                         # dX = dY' @ W needs a pack of W^T
        kind = "x3" if _x3_ok(M, N, Kp) else "wp"
        kind_t = "x3" if _x3_ok(M, Kp, N) else "wp"                    # dX: Kp outputs, N is the reduction dim
        wp = _packs(w, owner, kind, want_t and kind_t == kind)[0]
        wpt = _packs(w, owner, kind_t, True)[1] if want_t else None
        if tail is not None:
            tlink, t_src = tail
            if t_src:
                # first layer: K = 3 / 4 coordinates that need no gradient, output ReLU, direct publication of the weight gradient
                if (not ctx.needs_input_grad[0] and w_param is not None and Kp == 4 and relu_out and not relu_in
                        and res2 is None and grad_sum is None and not bw and N % 4 == 0):
                    tlink.armed = True
                    tlink.x4, tlink.w_param, tlink.b_param, tlink.k_orig = x2, w_param, b_param, K
                    tlink.fwd_key = _param_key(w_param)
                    ctx.set_materialize_grads(False)      # (a taken link: this node's backward receives None)
            elif (tlink.armed and want_t and not relu_in and not (bw & 2) and grad_sum is None and ctx.skip_dst is None
                  and Kp == K and N % 4 == 0 and M == tlink.x4.shape[0]):
                tlink.taken = True
                ctx.tail = tlink
        if lay and (kind != "x3" or init_gather is not None or bw or tail is not None or (lay == LAY_Y and res2 is not None)
                    or (lay == LAY_X and (grad_sum is not None or ctx.skip_dst is not None))):
            raise ValueError("G16 layout: a plain bf16x3 layer pair only (g16_pair_ok)")
        bits = None
        if init_gather is not None:
            if kind != "x3" or res2 is not None:
                raise ValueError("init_gather needs a layer on the bf16x3 kernel (gather_init_ok) without a residual")
            y = _fwd_x3_gather(x2, wp, N, b, init_gather, relu_in, relu_out)
        else:
            # (G16 output behind a ReLU, in a pass that will run backward: its mask as ReLU bits, written by this very launch)
            if (lay == LAY_Y and G16_BITS and relu_out and not ctx.premasked
                    and (ctx.needs_input_grad[0] or w_param is not None or ctx.needs_input_grad[1])):
                bits = relu_bits(M, N, x2.device)
            y = _run(kind, x2, wp, N, b, res2, None, None, relu_in, relu_out, res_sign, lay=lay, bits_out=bits)
        ctx.res_sign = res_sign
        ctx.relu_in, ctx.relu_out = relu_in, relu_out
        ctx.has_bias, ctx.has_res = b is not None, residual is not None
        ctx.x_shape, ctx.k_orig, ctx.n_out, ctx.kind_t = x.shape, K, N, kind_t
        ctx.fwd_key = _param_key(w_param) if w_param is not None else None
        # (the ReLU's mask for the backward pass: the bits where they were written, else the output itself)
        ctx.save_for_backward(x2, wpt, y if (relu_out and not ctx.premasked and bits is None) else None, bits)
        return y.reshape(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        if dy is None:          # (K4Tail: the next layer's dX GEMM produced this layer's weight gradient itself)
            return (None,) * 16
        x2, wpt, y, bits = ctx.saved_tensors
        N = ctx.n_out
        dy2 = dy.reshape(-1, N)
        dy2 = dy2 if dy2.is_contiguous() else dy2.contiguous()
        dx = dw = db = dres = None
        lay = ctx.lay
        if ctx.w_param is not None:
            fn = None
            if lay:      # LAY_Y: dY (and the mask: y, or its ReLU bits) arrive in G16; LAY_X: the input x2 is
                fn = _wgrad_g16_fn(1 if lay == LAY_Y else 2, bits)
            elif (REMASK_K4 and y is not None and x2.shape[1] == 4 and not ctx.relu_in and N % 4 == 0 and N >= 16
                    and x2.shape[0] >= 4096 and not ctx.has_res and ctx.fwd_key == _param_key(ctx.w_param)):
                # first layer of a position-encoding MLP: its ReLU mask is cheaper to recompute from the coordinates.
                # Only when the recomputed expression IS the forward one: no residual operand (the mask would be that of
                # relu(xW+b+res)), and the parameters are still the forward pass's (same pointer / version / epoch; a
                # changed key falls back to the saved output as the mask)
                fn = _wgrad_k4_remask(_padded_w4(ctx.w_param, batch=False), None if ctx.b_param is None else ctx.b_param.detach(),
                                      ctx.k_orig)
            if _use_side_stream(dy2):
                _wgrad_deferred(dy2, x2, y, ctx.relu_in, ctx.k_orig, ctx.w_param, ctx.b_param, fn=fn)   # side stream
            else:
                wgrad_direct(dy2, x2, y, ctx.relu_in, ctx.k_orig, ctx.w_param, ctx.b_param, fn=fn)
        elif ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            if lay:
                dw, db = _wgrad_g16_fn(1 if lay == LAY_Y else 2, bits)(dy2, x2, y, ctx.relu_in, ctx.has_bias)
            else:
                dw, db = _wgrad_sliced(dy2, x2, y, ctx.relu_in, ctx.has_bias, ctx.k_orig)
        tl = ctx.tail
        if tl is not None and ctx.needs_input_grad[0] and tl.fwd_key == _param_key(tl.w_param):
            # the dX GEMM with the K = 4 layer's weight gradient in its epilogue: no input gradient to report
            # -- and since nothing on the critical chain reads that gradient, the whole GEMM is weight-gradient work: side stream
            tfn = _k4tail_fn(tl, wpt, x2.shape[1], ctx.kind_t, x2)
            if K4_TAIL_SIDE and _use_side_stream(dy2):
                _wgrad_deferred(dy2, tl.x4, None, False, tl.k_orig, tl.w_param, tl.b_param, fn=tfn)
                wpt.record_stream(_side_stream(dy2.device))
            else:
                wgrad_direct(dy2, tl.x4, None, False, tl.k_orig, tl.w_param, tl.b_param, fn=tfn)
        elif ctx.needs_input_grad[0]:
            dyk, mk = dy2, y
            if N % 4:                                      # N = 3 (fc_out): pad the reduction dimension
                dyk = _pad_cols(dy2)
                mk = _pad_cols(y) if y is not None else None
            # dX = dY' @ W == linear(dY', W^T): W^T [K, N] as its fragment-major pack (rows >= K are zero)
            link = ctx.grad_sum
            addend = None
            if ctx.skip_dst is not None:
                addend, ctx.skip_dst.buf = ctx.skip_dst.buf, None
                if addend is not None and addend.shape[1] != x2.shape[1]:
                    raise RuntimeError("SkipGrad: the skip gradient does not have the layer's input width")
            # (G16: the gradient of a G16 output arrives in G16 -- this GEMM's X and mask; the gradient of a G16 input leaves in it)
            dx = _run(ctx.kind_t, dyk, wpt, x2.shape[1], None, link.buf if link is not None else None, mk,
                      x2 if (ctx.relu_in or ctx.mask_dx) else None, False, False, addend=addend,
                      lay=(LAY_X if lay == LAY_Y else LAY_Y if lay == LAY_X else 0), mask_bits=bits)
            dx = dx[:, :ctx.k_orig].reshape(ctx.x_shape) if ctx.k_orig != dx.shape[1] else dx.reshape(ctx.x_shape)
            if link is not None:         # running sum over the layers that share this input
                link.pending -= 1
                if link.pending > 0:
                    link.buf, dx = dx.reshape(-1, ctx.k_orig), None
                else:
                    link.buf, link.pending = None, link.total      # (re-armed for a second pass over a retained graph)
        if ctx.skip_src is not None:
            ctx.skip_src.buf = dy2            # (no output ReLU on this layer: d(residual) IS dy)
        elif ctx.has_res and ctx.needs_input_grad[3]:
            dres = dy2 if y is None else dy2 * (y > 0)
            dres = dres.reshape(dy.shape)
            if ctx.res_sign != 1.0:
                dres = -dres
        return dx, dw, db, dres, None, None, None, None, None, None, None, None, None, None, None, None


# Direct publication (`params=True`) hands weight gradients to `param.grad` behind autograd's back.  That is what makes
# the side stream possible, but it is invisible to everything that observes gradients THROUGH autograd: tensor hooks,
# post-accumulate-grad hooks (FSDP registers those), `torch.autograd.grad(loss, params)` and `backward(inputs=[...])`.
# Python-visible hooks are detected per call and switch that layer to the plain autograd path.  NOT detectable:
# torch's DistributedDataParallel -- its Reducer registers C++ post-hooks on the AccumulateGrad nodes, which Python
# cannot see.  Under DDP (and for autograd.grad / backward(inputs=)) wrap forward + backward in
# `with hip_linear.autograd_param_grads():` (or NSDP_PARAM_GRADS=autograd), or use nsdp_amd.parallel.GradAllReducer
# (the data-parallel path of this repo, which needs no hooks).
_PARAM_GRADS_DIRECT = os.environ.get("NSDP_PARAM_GRADS", "direct") != "autograd"


class autograd_param_grads:
    """Context manager: layers called inside return dW / db through autograd (main stream, no direct `.grad` writes)."""

    def __enter__(self):
        global _PARAM_GRADS_DIRECT
        self._was, _PARAM_GRADS_DIRECT = _PARAM_GRADS_DIRECT, False
        return self

    def __exit__(self, *exc):
        global _PARAM_GRADS_DIRECT
        _PARAM_GRADS_DIRECT = self._was
        return False


def _observed(t):
    """True when something watches this tensor's gradient through autograd (tensor hook, post-accumulate-grad hook)."""
    return bool(t._backward_hooks) or bool(getattr(t, "_post_accumulate_grad_hooks", None))


def linear(x, weight, bias=None, relu_in=False, relu_out=False, residual=None, params=False, grad_sum=None,
           out_f32=False, premasked=False, mask_dx=False, init_gather=None, residual_sign=1.0,
           skip_src=None, skip_dst=None, tail_src=None, tail_dst=None, pack_owner=None, lay=0):
    """``params=True``: `weight` / `bias` are the layer's leaf nn.Parameters; their gradients are then
    produced on the side stream and published to ``.grad`` at the end of the backward pass (see above).
    ``grad_sum``: an InputGradSum shared by the layers reading the same ``x``.
    ``out_f32``: with bf16 storage (nsdp_amd.precision) this layer's output stays fp32 (the network output).
    ``premasked`` / ``mask_dx`` (fp32 storage): the two halves of the Linear -> ReLU -> Linear backward contract, see
    _LinearFn.forward -- ``premasked`` on the layer with ``relu_out``, ``mask_dx`` on the ONLY reader of its output."""
    bw = (1 if premasked else 0) | (2 if mask_dx else 0)
    w_param = b_param = None
    if (params and _PARAM_GRADS_DIRECT and torch.is_grad_enabled() and weight.requires_grad and weight.is_leaf
            and not _observed(weight)):
        w_param = weight
        b_param = bias if (bias is not None and bias.requires_grad and bias.is_leaf) else None
        if bias is not None and (b_param is None or _observed(bias)):
            w_param = b_param = None   # mixed case: keep everything on the plain autograd path
    w2 = weight.squeeze(-1) if weight.dim() == 3 else weight   # 1x1 Conv1d weight [out, in, 1]
    # the pack cache lives on the layer's leaf parameter whether or not gradients are being recorded (eval / no_grad
    # forward passes would otherwise rebuild every pack at every call); staleness is covered by the cache key
    owner = weight if (params and weight.is_leaf) else None
    if pack_owner is not None:
        # a constant weight that is not a parameter (the fused decoder's zero-padded projection tables): `pack_owner` -- normally
        # the tensor itself -- carries the pack cache, so that the pack is built once instead of at every call
        owner = pack_owner
    if precision.is_bf16():
        if init_gather is not None or residual_sign != 1.0 or skip_src or skip_dst or lay:
            raise ValueError("init_gather / signed residuals / SkipGrad / G16 layouts belong to fp32 storage")
        if bw:
            raise ValueError("premasked / mask_dx belong to fp32 storage (bf16 storage uses relu_in on the second layer)")
        from . import hip_linear_bf16 as hb
        K, N = x.shape[-1], w2.shape[0]
        if (hb.NATIVE and x.dtype is torch.bfloat16 and 8 <= K <= 256 and K % 8 == 0 and N <= 256
                and (out_f32 or N % 4 == 0)):
            if residual is not None and residual.dtype is not torch.bfloat16:
                residual = residual.to(torch.bfloat16)
            return hb.linear(x, weight, bias, relu_in, relu_out, residual, w_param, b_param, grad_sum, owner, out_f32)
        if hb.NATIVE and not out_f32 and grad_sum is None and hb.k4_supported(x, N, relu_in, residual):
            return hb.linear_k4(x, weight, bias, relu_out, w_param, b_param)       # fp32 coordinates in, bf16 out
        # the other narrow ends of the network (enc_sdf's 4 / 7 input features): fp32 kernels, result rounded to bf16
        if grad_sum is not None and hb.NATIVE:
            raise ValueError("InputGradSum needs a bf16 layer in bf16 storage mode")
        xf = x if x.dtype is torch.float32 else x.float()
        rf = residual if (residual is None or residual.dtype is torch.float32) else residual.float()
        if w_param is not None:
            y = _LinearFn.apply(xf, w2.detach(), None if bias is None else bias.detach(), rf, bool(relu_in),
                                bool(relu_out), w_param, b_param, grad_sum)
        else:
            y = _LinearFn.apply(xf, w2, bias, rf, bool(relu_in), bool(relu_out), None, None, grad_sum, owner)
        return y if out_f32 else y.to(torch.bfloat16)
    if skip_src is not None and skip_dst is not None:
        raise ValueError("a layer is either end of a SkipGrad link")
    skip = (skip_src, True) if skip_src is not None else (skip_dst, False) if skip_dst is not None else None
    tail = (tail_src, True) if tail_src is not None else (tail_dst, False) if tail_dst is not None else None
    if w_param is not None:
        # the Function sees detached operands for the weights: their gradient does not go through autograd
        return _LinearFn.apply(x, w2.detach(), None if bias is None else bias.detach(), residual, bool(relu_in),
                               bool(relu_out), w_param, b_param, grad_sum, None, bw, init_gather,
                               float(residual_sign), skip, tail, int(lay))
    return _LinearFn.apply(x, w2, bias, residual, bool(relu_in), bool(relu_out), None, None, grad_sum, owner, bw,
                           init_gather, float(residual_sign), skip, None, int(lay))
