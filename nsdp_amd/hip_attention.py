"""Autograd wrappers of the fused vector-attention glue kernels (csrc/attention.hip)."""
from __future__ import annotations

import ctypes
import os

import torch

from ._lib import check, fptr, iptr, lib, on_device, optptr, stream_ptr

_ci = ctypes.c_int


BF16 = torch.bfloat16


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def _p(t, dtype, name="tensor"):
    """Device pointer of an activation tensor of the storage type of this call (fp32 or bf16), or NULL."""
    if t is None:
        return ctypes.c_void_p(0)
    if t.dtype is not dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype} (all activation tensors of one call share a storage type)")
    if not t.is_cuda or not t.is_contiguous():
        raise TypeError(f"{name} must be a contiguous GPU tensor")
    return ctypes.c_void_p(t.data_ptr())


def _fn(name, dtype):
    return getattr(lib(), name + ("_bf16" if dtype is BF16 else ""))


# Decoder (<= 128 table rows per shape, tens of thousands of rows to scatter): the anchor-table gradients as a GEMM against
# the one-hot index matrix -- no atomics, a fixed summation order (the train step is bit-reproducible with it), and faster
# than the register-table / LDS-table atomic kernels.  bf16 storage: nsdp_scatter_rows_onehot_bf16; fp32: the three bf16
# planes of the source times the (exact) one-hot operand, nsdp_scatter_rows_onehot_f32.  "0": the atomic kernels (A/B knob).
ONEHOT_SCATTER = os.environ.get("NSDP_ONEHOT_SCATTER", "1") != "0"
ONEHOT_SCATTER_F32 = os.environ.get("NSDP_ONEHOT_SCATTER_F32", "1") != "0"


def _onehot_ok(dt, qb_or_decoder, N, d):
    if not qb_or_decoder:
        return False
    if dt is BF16:
        return ONEHOT_SCATTER and N <= 128 and N % 2 == 0 and d % 8 == 0 and d <= 256
    return ONEHOT_SCATTER_F32 and dt is torch.float32 and N <= 128 and d % 4 == 0 and 16 < d <= 208


# Encoder levels whose source table fits neither LDS nor registers (2048 / 500 source points): scatter through inverse
# neighbour lists (csrc/segment.hip) instead of global fp32 atomics.  The lists depend on the index tensor only and are
# cached on it (the two attentions of a set abstraction share one index set; forward builds nothing).
# the two small corrections of the fused d(pos) hand-over (dkf += dvf, dq -= dy) inside the scatter / reduction kernels ("0": as two
# elementwise launches behind them, A/B knob)
FOLD_PRE_BWD = os.environ.get("NSDP_FOLD_PRE_BWD", "1") != "0"
INVERSE_LISTS = os.environ.get("NSDP_INVERSE_LISTS", "1")         # "0": the global-atomic kernels (A/B knob)


def _use_inverse(dt, qb, n, N, d):
    # (the decoder -- one query vector per shape, 57 344 entries over 100 anchors -- keeps its register table / scatter as a
    # GEMM: its lists are built by ONE workgroup per shape, 2 ms per step, more than the 0.4 ms the segment sum would save)
    if qb or N > 8192 or INVERSE_LISTS == "0":
        return False
    table_fits_lds = N * d * 4 <= 110 * 1024 and n >= 4 * N            # (lds_table_fits of csrc/attention.hip)
    if table_fits_lds:
        return False
    return True


def inverse_lists(idx, N):
    """(offsets [B,N+1], entries [B,E]) of idx [B,n,k] (nsdp_knn_invert), cached on the index tensor.
    The cache is keyed by the CONTENTS' identity as far as PyTorch exposes it -- (N, storage pointer, version counter): an
    index buffer that is refilled in place (`copy_` into a static input of a captured step) gets new lists.  While a stream
    capture is running the cache is neither read nor written: the list build must be a node of the graph (a replay sees
    new index contents at the same address and version), and lists built for a capture must not outlive it."""
    capturing = idx.is_cuda and torch.cuda.is_current_stream_capturing()
    if os.environ.get("NSDP_GEOMETRY_ABLATE", "0") == "1":      # (timing-only ablation, see model/ops.py: lists outlive the capture)
        capturing = False
    cache = idx.__dict__.setdefault("_nsdp_inverse", {}) if (hasattr(idx, "__dict__") and not capturing) else {}
    key = (N, idx.data_ptr(), idx._version)
    hit = cache.get(key)
    if hit is not None:
        return hit
    for stale in [k for k in cache if k[1:] != key[1:]]:      # (lists of older contents of this buffer)
        del cache[stale]
    B = idx.shape[0]
    E = idx.numel() // B
    offsets = torch.empty((B, N + 1), dtype=torch.int32, device=idx.device)
    entries = torch.empty((B, E), dtype=torch.int32, device=idx.device)
    with on_device(idx):
        check(lib().nsdp_knn_invert(iptr(idx, "idx"), _ci(B), _ci(E), _ci(N), iptr(offsets), iptr(entries), stream_ptr()),
              "nsdp_knn_invert")
    cache[key] = (offsets, entries)
    return offsets, entries


def segment_sum(src, idx, N, scale=1.0, inv=None, addend=None):
    """out [B,N,d] fp32 = scale * scatter-add of the rows of src [B,n,k,d] by idx [B,n,k] (+ ``addend`` [B,N,d] fp32), as a
    gather-reduce over the inverse lists (deterministic, no atomics).  ``inv``: the lists, when the caller built them already."""
    B = src.shape[0]
    d = src.shape[-1]
    E = idx.numel() // B
    offsets, entries = inv if inv is not None else inverse_lists(idx, N)
    out = torch.empty((B, N, d), dtype=torch.float32, device=src.device)
    dt = src.dtype
    with on_device(src):
        if addend is not None:
            check(_fn("nsdp_segment_sum_rows_add", dt)(_p(src, dt, "src"), iptr(offsets), iptr(entries), _ci(B), _ci(E), _ci(N),
                                                       _ci(d), ctypes.c_float(scale), fptr(addend, "addend"), fptr(out), stream_ptr()),
                  "nsdp_segment_sum_rows_add")
        else:
            check(_fn("nsdp_segment_sum_rows", dt)(_p(src, dt, "src"), iptr(offsets), iptr(entries), _ci(B), _ci(E), _ci(N), _ci(d),
                                                   ctypes.c_float(scale), fptr(out), stream_ptr()), "nsdp_segment_sum_rows")
    return out


def onehot_scatter(src, idx, N):
    """table [B,N,d] fp32 = sum of the rows of src [B,rows,d] (bf16 / fp32) by idx [B,rows] (nsdp_scatter_rows_onehot_*)."""
    B, rows, d = src.shape
    L = lib()
    if src.dtype is torch.float32:
        L.nsdp_scatter_rows_onehot_f32_workspace_bytes.restype = ctypes.c_size_t
        nbytes = int(L.nsdp_scatter_rows_onehot_f32_workspace_bytes(_ci(B), ctypes.c_longlong(rows), _ci(N), _ci(d)))
        ws = torch.empty(max(nbytes // 4, 1), dtype=torch.float32, device=src.device)
        table = torch.empty((B, N, d), dtype=torch.float32, device=src.device)
        with on_device(src):
            check(L.nsdp_scatter_rows_onehot_f32(fptr(src, "src"), iptr(idx, "idx"), _ci(B), ctypes.c_longlong(rows), _ci(N),
                                                 _ci(d), fptr(table), fptr(ws), ctypes.c_size_t(nbytes), stream_ptr()),
                  "nsdp_scatter_rows_onehot_f32")
        return table
    L.nsdp_scatter_rows_onehot_bf16_workspace_bytes.restype = ctypes.c_size_t
    nbytes = int(L.nsdp_scatter_rows_onehot_bf16_workspace_bytes(_ci(B), ctypes.c_longlong(rows), _ci(N), _ci(d)))
    ws = torch.empty(max(nbytes // 4, 1), dtype=torch.float32, device=src.device)
    table = torch.empty((B, N, d), dtype=torch.float32, device=src.device)
    with on_device(src):
        check(L.nsdp_scatter_rows_onehot_bf16(_p(src, BF16, "src"), iptr(idx, "idx"), _ci(B), ctypes.c_longlong(rows), _ci(N),
                                              _ci(d), fptr(table), fptr(ws), ctypes.c_size_t(nbytes), stream_ptr()),
              "nsdp_scatter_rows_onehot_bf16")
    return table


class _PosGrad:
    """Hand-over of d(pos) between the two halves of one attention block.  `pos` feeds the logits (attn_pre) and the
    values (attn_post); in the backward pass attn_post runs first (attn_pre's gradient depends on it through the
    gamma MLP), parks its d(pos) here and reports no gradient for `pos`.

    With ``grad_sum`` (a hip_linear.InputGradSum shared with the FIRST layer of the gamma MLP, the only consumer of `u`)
    the sum d(pos) = d(u) + d(pos)|values costs no pass of its own: attn_post's d(pos) becomes the `residual` operand of
    that layer's dX GEMM, which therefore produces the TOTAL and hands it to attn_pre's backward as "d(u)".  attn_pre
    then only scatters (one read of the [B,n,k,d] tensor instead of a read-modify-write of a second one) and corrects
    by linearity:  scatter(d(u)) = scatter(total) - scatter(d(pos)|values) = scatter(total) - dvf,  and the row sums the
    same way (sum_j d(pos)|values_ij = dy_i for a softmax over the neighbours; sum over a shape = sum_a dvf[a]).
    Without ``grad_sum`` attn_pre's kernel adds d(u) into the parked tensor (the round-1 form)."""
    __slots__ = ("dpos", "grad_sum", "dvf", "dy", "fused", "qb")

    def __init__(self):
        self.dpos = None
        self.grad_sum = None      # InputGradSum of the gamma MLP's first layer (set by ops.vector_attention)
        self.dvf = None           # fp32 scatter of attn_post's d(pos) = its dvf
        self.dy = None            # upstream gradient of attn_post (per-point correction of dq)
        self.fused = False        # attn_post's backward has run and primed grad_sum.buf
        self.qb = False           # one query vector per shape (set by attn_pre's forward)


class _AttnPre(torch.autograd.Function):
    """u = q[:, :, None] - kf[idx] + pos."""

    @staticmethod
    def forward(ctx, q, kf, pos, idx, link=None, inv=None, pre=None):
        ctx.link = link
        ctx.inv = inv
        q, kf, pos = _c(q), _c(kf), _c(pos)
        B, n, k, d = pos.shape
        N = kf.shape[1]
        qb = int(q.shape[1] == 1 and n != 1)  # (B,1,d): one query vector per shape
        if link is not None:
            link.qb = bool(qb)
        ctx.save_for_backward(idx)
        ctx.dims = (B, n, N, k, d, qb)
        if pre is not None:          # computed by a fused forward kernel: this node only records the backward
            return pre.reshape(pos.shape)
        u = torch.empty_like(pos)
        dt = pos.dtype
        with on_device(pos):
            check(_fn("nsdp_attn_pre_fwd", dt)(_p(q, dt, "q"), _p(kf, dt, "kf"), _p(pos, dt, "pos"), iptr(idx, "idx"), _ci(B),
                                               _ci(n), _ci(N), _ci(k), _ci(d), _ci(qb), _p(u, dt), stream_ptr()),
                  "nsdp_attn_pre_fwd")
        return u

    @staticmethod
    def backward(ctx, du):
        (idx,) = ctx.saved_tensors
        B, n, N, k, d, qb = ctx.dims
        du = _c(du)
        dt = du.dtype
        dq = torch.empty((B, 1 if qb else n, d), dtype=torch.float32, device=du.device)
        dkf = torch.empty((B, N, d), dtype=torch.float32, device=du.device)
        acc = None
        link = ctx.link
        fused = link is not None and link.fused
        if link is not None and not fused:
            acc, link.dpos = link.dpos, None
        if acc is None and _use_inverse(dt, qb, n, N, d):
            # dq from a pure stream over du, dkf = -scatter(du) as a gather-reduce over the inverse neighbour lists
            fold = fused and not qb and FOLD_PRE_BWD and link.dvf.dtype is torch.float32 and link.dy.dtype is dt
            with on_device(du):
                if fold:      # (... with the two corrections of the fused d(pos) hand-over below folded into the kernels)
                    check(_fn("nsdp_attn_pre_bwd_sub", dt)(_p(du, dt, "du"), iptr(idx), _ci(B), _ci(n), _ci(N), _ci(k), _ci(d),
                                                           _p(_c(link.dy), dt, "dy"), fptr(dq), stream_ptr()), "nsdp_attn_pre_bwd_sub")
                else:
                    check(_fn("nsdp_attn_pre_bwd", dt)(_p(du, dt, "du"), iptr(idx), _ci(B), _ci(n), _ci(N), _ci(k), _ci(d),
                                                       _ci(qb), fptr(dq), ctypes.c_void_p(0), ctypes.c_void_p(0), stream_ptr()),
                          "nsdp_attn_pre_bwd")
            dkf = segment_sum(du, idx, N, -1.0, ctx.inv, addend=_c(link.dvf) if fold else None)
            if fold:
                link.dvf = link.dy = None
                link.fused = fused = False
        elif fused and acc is None and _onehot_ok(dt, qb, N, d):
            # decoder, bf16: -scatter(du) and the per-shape sum of du from one scatter-as-GEMM pass (no atomics)
            table = onehot_scatter(du.reshape(B, n * k, d), idx.reshape(B, n * k), N)
            dkf = table.neg_()
            dq = dkf.sum(1, keepdim=True).neg_()
        else:
            with on_device(du):
                check(_fn("nsdp_attn_pre_bwd", dt)(_p(du, dt, "du"), iptr(idx), _ci(B), _ci(n), _ci(N), _ci(k), _ci(d),
                                                   _ci(qb), fptr(dq), fptr(dkf), _p(acc, dt, "dpos"), stream_ptr()),
                      "nsdp_attn_pre_bwd")
        if fused:
            # `du` is the total d(pos) (see _PosGrad): undo the value path's share in the two small outputs
            dkf.add_(link.dvf)
            if qb:
                dq.sub_(link.dvf.sum(1, keepdim=True))
            else:
                dq.sub_(link.dy.reshape(dq.shape))
            link.dvf = link.dy = None
            link.fused = False
        if dt is BF16:           # (scatter / reduction outputs are produced in fp32; the tables are small)
            dq, dkf = dq.to(BF16), dkf.to(BF16)
        return dq, dkf, (du if acc is None else acc), None, None, None, None


class _AttnPost(torch.autograd.Function):
    """y = sum_j softmax_j(a) * (vf[idx] + pos) [+ global token] [+ residual]."""

    @staticmethod
    def forward(ctx, a, vf, pos, idx, a_g, v_g, residual, link=None, inv=None, sub=None):
        # sub = (kf, q), constants of this node: `pos` holds u = q_i - k_j + pos (hip_linear's init_gather: pos itself was never
        # materialised).  The values v_j + pos_ij are then u + (v + k)[idx] - q_i: the kernels get the table v + k (minus q
        # when it is one vector per shape) in place of vf and, for per-point queries, q as `qsub`.  Gradients are the original
        # graph's: d(pos) = w dy, d(vf) = its scatter, nothing for k / q from this node.
        ctx.link = link
        ctx.inv = inv
        a, pos = _c(a), _c(pos)
        qsub = None
        if sub is not None:
            kf_c, q_c = sub
            if vf is None or a.dtype is not torch.float32:
                raise ValueError("attn_post(sub=): needs a value table and fp32 storage")
            per_shape = q_c.shape[1] == 1 and a.shape[1] != 1
            if kf_c is None:                      # the caller's table already holds v + k (ops.vector_attention, combined=)
                if per_shape:
                    raise ValueError("attn_post(sub=(None, q)): per-point queries only")
            else:
                vf = vf + kf_c - q_c if per_shape else vf + kf_c
            qsub = None if per_shape else _c(q_c)
            if qsub is not None and a_g is not None:
                raise ValueError("attn_post(sub=): per-point queries and a global token do not combine")
        vf = None if vf is None else _c(vf)
        a_g = None if a_g is None else _c(a_g)
        v_g = None if v_g is None else _c(v_g)
        residual = None if residual is None else _c(residual)
        B, n, k, d = a.shape
        N = vf.shape[1] if vf is not None else 1
        dt = a.dtype
        y = torch.empty((B, n, d), dtype=dt, device=a.device)
        lse = torch.empty((B, n, d), dtype=torch.float32, device=a.device)
        with on_device(a):
            if qsub is not None:
                check(lib().nsdp_attn_post_fwd_q(fptr(a, "a"), fptr(vf, "vk"), fptr(pos, "u"), iptr(idx, "idx"), fptr(qsub, "q"),
                                                 optptr(residual), _ci(B), _ci(n), _ci(N), _ci(k), _ci(d), fptr(y), fptr(lse),
                                                 stream_ptr()), "nsdp_attn_post_fwd_q")
            else:
                check(_fn("nsdp_attn_post_fwd", dt)(_p(a, dt, "a"), _p(vf, dt, "vf"), _p(pos, dt, "pos"), iptr(idx, "idx"),
                                                    _p(a_g, dt, "a_g"), _p(v_g, dt, "v_g"), _p(residual, dt, "residual"),
                                                    _ci(B), _ci(n), _ci(N), _ci(k), _ci(d), _p(y, dt), fptr(lse),
                                                    stream_ptr()), "nsdp_attn_post_fwd")
        ctx.save_for_backward(a, vf, pos, idx, a_g, v_g, y, residual, lse, qsub)
        ctx.dims = (B, n, N, k, d)
        return y

    @staticmethod
    def backward(ctx, dy):
        a, vf, pos, idx, a_g, v_g, y, residual, lse, qsub = ctx.saved_tensors
        B, n, N, k, d = ctx.dims
        dy = _c(dy)
        dev = dy.device
        dt = a.dtype
        da = torch.empty_like(a)
        dpos = torch.empty_like(a)
        onehot = vf is not None and _onehot_ok(dt, a_g is not None, N, d)
        link = ctx.link
        qb = bool(link.qb) if link is not None else False
        # (a per-shape-query block has no inverse lists -- backward_lists(qb=True) builds none -- and must not build them here)
        inverse = (vf is not None and not onehot and a_g is None and _use_inverse(dt, qb, n, N, d)
                   and (ctx.inv is not None or not qb))
        dvf = torch.empty((B, N, d), dtype=torch.float32, device=dev) if (vf is not None and not onehot and not inverse) else None
        da_g = torch.empty((B, d), dtype=torch.float32, device=dev) if a_g is not None else None
        dv_g = torch.empty((B, d), dtype=torch.float32, device=dev) if a_g is not None else None
        if onehot:
            # pure stream, no atomics at all (the global-token sums go through per-workgroup partials in a fixed order)
            L = lib()
            L.nsdp_attn_post_bwd_det_workspace_bytes.restype = ctypes.c_size_t
            nbytes = int(L.nsdp_attn_post_bwd_det_workspace_bytes(_ci(B), _ci(n), _ci(k), _ci(d))) if a_g is not None else 0
            ws = torch.empty(max(nbytes // 4, 1), dtype=torch.float32, device=dev)
            with on_device(dy):
                check(_fn("nsdp_attn_post_bwd_det", dt)(_p(dy, dt, "dy"), _p(a, dt), _p(vf, dt), _p(pos, dt), iptr(idx),
                                                        _p(a_g, dt), _p(v_g, dt), _p(y, dt), _p(residual, dt), fptr(lse), _ci(B),
                                                        _ci(n), _ci(N), _ci(k), _ci(d), _p(da, dt), _p(dpos, dt), optptr(da_g),
                                                        optptr(dv_g), fptr(ws), ctypes.c_size_t(nbytes), stream_ptr()),
                      "nsdp_attn_post_bwd_det")
        elif qsub is not None:
            with on_device(dy):
                check(lib().nsdp_attn_post_bwd_q(fptr(dy, "dy"), fptr(a), fptr(vf), fptr(pos), iptr(idx), fptr(qsub), fptr(y),
                                                 optptr(residual), fptr(lse), _ci(B), _ci(n), _ci(N), _ci(k), _ci(d), fptr(da),
                                                 fptr(dpos), optptr(dvf), stream_ptr()), "nsdp_attn_post_bwd_q")
        else:
            with on_device(dy):
                check(_fn("nsdp_attn_post_bwd", dt)(_p(dy, dt, "dy"), _p(a, dt), _p(vf, dt), _p(pos, dt), iptr(idx),
                                                    _p(a_g, dt), _p(v_g, dt), _p(y, dt), _p(residual, dt), fptr(lse), _ci(B),
                                                    _ci(n), _ci(N), _ci(k), _ci(d), _p(da, dt), _p(dpos, dt), optptr(dvf),
                                                    optptr(da_g), optptr(dv_g), stream_ptr()), "nsdp_attn_post_bwd")
        if onehot:     # the kernel only streamed; dvf = scatter(d(pos)) as a GEMM against the one-hot index matrix
            dvf = onehot_scatter(dpos.reshape(B, n * k, d), idx.reshape(B, n * k), N)
        elif inverse:  # ... or as a gather-reduce over the inverse neighbour lists
            dvf = segment_sum(dpos, idx, N, 1.0, ctx.inv)
        # the fused d(pos) hand-over corrects dq by `- dy`, i.e. relies on sum_j softmax_ij = 1 over the NEIGHBOURS: with a
        # global token the softmax has k + 1 entries and that only holds per shape (the qb form corrects by - sum_a dvf[a])
        if (link is not None and link.grad_sum is not None and ctx.needs_input_grad[2] and dvf is not None
                and (a_g is None or qb)):
            link.grad_sum.buf = dpos.reshape(-1, d)         # residual of the gamma MLP's first dX GEMM
            link.dvf, link.dy, link.fused = dvf, dy, True
        if dt is BF16:
            dvf = None if dvf is None else dvf.to(BF16)
            da_g = None if da_g is None else da_g.to(BF16)
            dv_g = None if dv_g is None else dv_g.to(BF16)
        if link is not None and link.fused:
            dpos = None                               # travels as the dX GEMM's residual; attn_pre reports the total
        elif link is not None and ctx.needs_input_grad[2]:
            link.dpos, dpos = dpos, None              # attn_pre's backward adds d(u) and reports the sum
        return da, dvf, dpos, None, da_g, dv_g, (dy if residual is not None else None), None, None, None


def pos_grad_link():
    """One per attention block whose `pos` goes through both attn_pre and attn_post (and whose attn_post output
    depends on the attn_pre output, which is what orders the two backward calls): pass it to both."""
    return _PosGrad()


NATIVE_BF16 = True        # bf16-storage kernels (False: cast around the fp32 ones -- the reference semantics the tests compare with)


def _f(t):
    return None if t is None else (t if t.dtype is torch.float32 else t.float())


def native(t):
    """Does this tensor's storage type go to the kernels directly (not through the cast-reference path)?"""
    return t.dtype is torch.float32 or NATIVE_BF16


def backward_lists(idx, n, N, d, qb=False):
    """The inverse neighbour lists the backward pass of an attention block over this index set will want, or None.
    Built in the FORWARD pass (once per index set: the cache lives on the index tensor object, which the backward pass
    no longer sees) and handed to attn_pre / attn_post as ``inv``."""
    if not torch.is_grad_enabled() or not _use_inverse(torch.float32, qb, n, N, d):
        return None
    return inverse_lists(idx, N)


def attn_pre(q, kf, pos, idx, link=None, inv=None, precomputed=None):
    if pos.dtype is torch.bfloat16 and not NATIVE_BF16:
        return _AttnPre.apply(_f(q), _f(kf), _f(pos), idx, link, inv).to(torch.bfloat16)
    return _AttnPre.apply(q, kf, pos, idx, link, inv, precomputed)


def attn_post(a, vf, pos, idx, a_g=None, v_g=None, residual=None, link=None, inv=None, sub=None):
    if a.dtype is torch.bfloat16 and not NATIVE_BF16:
        return _AttnPost.apply(_f(a), _f(vf), _f(pos), idx, _f(a_g), _f(v_g), _f(residual), link, inv).to(torch.bfloat16)
    return _AttnPost.apply(a, vf, pos, idx, a_g, v_g, residual, link, inv, sub)
