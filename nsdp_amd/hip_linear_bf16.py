"""Dense layers with bf16 STORAGE (BASELINE config 3): autograd wrapper over nsdp_linear_bf16 / nsdp_linear_wgrad_bf16
(csrc/gemm_bf16.hip).  Activations, residuals, ReLU masks and gradients of activations are bf16 tensors; weights,
biases and their gradients are fp32 (master weights); every product accumulates in fp32 on the matrix cores.

Layers the bf16 kernels do not cover (a reduction or output width that is not a multiple of 8 / 4: the 3-wide
coordinate inputs, fc_out's 3 outputs in the backward direction) go through the fp32 kernels of hip_linear on cast
copies -- they are the small ends of the network."""
from __future__ import annotations

import ctypes

import torch

from . import hip_linear
from ._lib import check, fptr, hptr, lib, on_device, opthptr, optptr, stream_ptr

_ll = ctypes.c_longlong
_ci = ctypes.c_int
BF16 = torch.bfloat16
NATIVE = True     # False: every layer through the fp32 kernels on casts, results rounded to bf16 (the reference semantics
                  # of bf16 storage that tests compare the native kernels with)


def supported(N, K):
    """Shape contract of nsdp_linear_bf16 with bf16 output (reduction width K, output width N)."""
    return 8 <= K <= 256 and K % 8 == 0 and 4 <= N <= 256 and N % 4 == 0


def pack_weight_b16(w, fwd=True, transposed=False):
    """bf16 fragment packs of W [N,K] fp32 (nsdp_pack_weights_bf16): (Wp or None, WpT or None), uint8 buffers."""
    N, K = w.shape
    L = lib()
    L.nsdp_packed_weight_bf16_bytes.restype = ctypes.c_longlong
    wp = wpt = None
    if fwd:
        wp = torch.empty(int(L.nsdp_packed_weight_bf16_bytes(_ci(N), _ci(K), _ci(0))), dtype=torch.uint8, device=w.device)
    if transposed:
        wpt = torch.empty(int(L.nsdp_packed_weight_bf16_bytes(_ci(N), _ci(K), _ci(1))), dtype=torch.uint8, device=w.device)
    d = (hip_linear._PackDesc * 1)()
    d[0].W, d[0].N, d[0].K, d[0].kind = w.data_ptr(), N, K, 2
    d[0].Wp = wp.data_ptr() if wp is not None else None
    d[0].WpT = wpt.data_ptr() if wpt is not None else None
    with on_device(w):
        fptr(w, "weight")
        check(L.nsdp_pack_weights_bf16(d, _ci(1), stream_ptr()), "nsdp_pack_weights_bf16")
    return wp, wpt


def run(x2, pack, N, b, residual, mask, out_mask, relu_in, relu_out, out_f32=False):
    """Y = post(pre(x2) W^T + b (+ residual)) with W given as its bf16 pack (logical [N, K = x2.shape[1]])."""
    M, K = x2.shape
    y = torch.empty((M, N), dtype=torch.float32 if out_f32 else BF16, device=x2.device)
    with on_device(x2):
        check(lib().nsdp_linear_bf16(hptr(x2, "x"), ctypes.c_void_p(pack.data_ptr()), optptr(b), opthptr(residual, "residual"),
                                     opthptr(mask, "mask"), opthptr(out_mask, "out_mask"), ctypes.c_void_p(y.data_ptr()),
                                     _ll(M), _ci(N), _ci(K), _ci(int(relu_in)), _ci(int(relu_out)), _ci(int(out_f32)),
                                     stream_ptr()), "nsdp_linear_bf16")
    return y


def wgrad(dy2, x2, mask, relu_x, want_db, out=None):
    """dW [N,K], db [N] (fp32) from bf16 dY [M,N], X [M,K] (nsdp_linear_wgrad_bf16); `out`: accumulate into these."""
    M, N = dy2.shape
    K = x2.shape[1]
    L = lib()
    if mask is not None and not L.nsdp_linear_wgrad_bf16_takes_mask(_ll(M), _ci(N), _ci(K)):
        dy2, mask = dy2 * (mask > 0), None            # (the mask rows do not fit into the kernel's LDS ring as well)
    L.nsdp_linear_wgrad_bf16_workspace_bytes.restype = ctypes.c_size_t
    nbytes = int(L.nsdp_linear_wgrad_bf16_workspace_bytes(_ll(M), _ci(N), _ci(K)))
    ws = torch.empty(max(nbytes // 4, 1), dtype=torch.float32, device=dy2.device)
    dw, db, acc = hip_linear.wgrad_out(out, N, K, want_db, dy2.device)
    batch = hip_linear._cur_reduce
    if batch is not None and hip_linear.BATCH_REDUCE > 0:
        # the row kernel now, the reduction with the pass's batch (see hip_linear._wgrad_x3: same protocol)
        ptrs = {dw.data_ptr()} | ({db.data_ptr()} if db is not None else set())
        with on_device(dy2):
            if ptrs & batch["targets"]:
                hip_linear._flush_reduce(batch)
            desc = hip_linear._ReduceDescB16()
            check(L.nsdp_linear_wgrad_bf16_partials(hptr(dy2, "dy"), hptr(x2, "x"), opthptr(mask, "mask"), _ci(int(relu_x)),
                                                    fptr(dw), optptr(db), _ll(M), _ci(N), _ci(K), _ci(acc), fptr(ws),
                                                    ctypes.c_size_t(nbytes), ctypes.byref(desc), stream_ptr()),
                  "nsdp_linear_wgrad_bf16_partials")
            if desc.ws:
                batch["descs_b16"].append(desc)
                batch["keep"].append((ws, dw, db, dy2, x2, mask))
                batch["targets"] |= ptrs
                if len(batch["descs"]) + len(batch["descs_b16"]) >= hip_linear.BATCH_REDUCE:
                    hip_linear._flush_reduce(batch)
        return dw, db
    with on_device(dy2):
        check(L.nsdp_linear_wgrad_bf16(hptr(dy2, "dy"), hptr(x2, "x"), opthptr(mask, "mask"), _ci(int(relu_x)), fptr(dw),
                                       optptr(db), _ll(M), _ci(N), _ci(K), _ci(acc), fptr(ws), ctypes.c_size_t(nbytes),
                                       stream_ptr()), "nsdp_linear_wgrad_bf16")
    return dw, db


def _wgrad_any(dy2, x2, y_mask, relu_x, want_db, out=None):
    """Weight gradient of a bf16-storage layer; shapes outside the bf16 kernel go through the fp32 kernels on casts."""
    N, K = dy2.shape[1], x2.shape[1]
    if dy2.dtype is BF16 and N % 2 == 0 and K % 2 == 0 and N <= 256 and K <= 256 and N >= 8 and K >= 8:
        return wgrad(dy2, x2, y_mask, relu_x, want_db, out)
    dyf, xf = dy2.float(), x2.float()
    mk = None if y_mask is None else y_mask.float()
    kp = (-K) % 4
    if kp:
        xf = torch.nn.functional.pad(xf, (0, kp))
    prev, hip_linear._cur_reduce = hip_linear._cur_reduce, None      # (the slice below reads dw at once: no pending reduction)
    try:
        dw, db = hip_linear._wgrad(dyf.contiguous(), xf.contiguous(), mk, relu_x, want_db, None if kp else out)
    finally:
        hip_linear._cur_reduce = prev
    return (dw[:, :K].contiguous() if kp else dw), db


_wgrad_any.batched_reduce = True      # (hip_linear._wgrad_deferred: this routine hands its reduction to the pass's batch)


class _LinearB16Fn(torch.autograd.Function):
    """bf16-storage counterpart of hip_linear._LinearFn (same fused prologues / epilogues, same side-stream and
    direct-publication protocol for the parameter gradients, same InputGradSum hand-over)."""

    @staticmethod
    def forward(ctx, x, w, b, residual, relu_in, relu_out, w_param, b_param, grad_sum, owner, out_f32):
        ctx.w_param, ctx.b_param = w_param, b_param
        owner = w_param if w_param is not None else owner
        ctx.grad_sum = None
        if grad_sum is not None and ctx.needs_input_grad[0]:
            if relu_in:
                raise ValueError("InputGradSum: layers with a fused input ReLU cannot join")
            grad_sum.total += 1
            grad_sum.pending += 1
            ctx.grad_sum = grad_sum
        K = x.shape[-1]
        N = w.shape[0]
        x2 = x.reshape(-1, K)
        x2 = x2 if x2.is_contiguous() else x2.contiguous()
        res2 = None
        if residual is not None:
            res2 = residual.reshape(-1, N)
            res2 = res2 if res2.is_contiguous() else res2.contiguous()
        want_t = bool(ctx.needs_input_grad[0])
        t_ok = supported(K, N)                       # dX = dY W: K outputs, reduction over N
        wp = hip_linear._packs(w, owner, "b16", want_t and t_ok)[0]
        wpt = hip_linear._packs(w, owner, "b16", True)[1] if (want_t and t_ok) else None
        y = run(x2, wp, N, b, res2, None, None, relu_in, relu_out, out_f32)
        ctx.relu_in, ctx.relu_out = relu_in, relu_out
        ctx.has_bias, ctx.has_res = b is not None, residual is not None
        ctx.x_shape, ctx.n_out, ctx.t_ok = x.shape, N, t_ok
        # (the fp32 fallback of the backward pass -- taken whenever dY arrives in fp32, i.e. for every out_f32 layer --
        # multiplies by the plain weight: keep it for those as well as for the shapes without a W^T pack)
        ctx.save_for_backward(x2, wpt, y if relu_out else None, w if (out_f32 or not t_ok) else None)
        return y.reshape(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, wpt, y, w_plain = ctx.saved_tensors
        N = ctx.n_out
        K = x2.shape[1]
        dy2 = dy.reshape(-1, N)
        dy2 = dy2 if dy2.is_contiguous() else dy2.contiguous()
        dx = dw = db = dres = None
        if ctx.w_param is not None:
            if hip_linear._use_side_stream(dy2):
                hip_linear._wgrad_deferred(dy2, x2, y, ctx.relu_in, K, ctx.w_param, ctx.b_param, fn=_wgrad_any)
            else:
                hip_linear.wgrad_direct(dy2, x2, y, ctx.relu_in, K, ctx.w_param, ctx.b_param, fn=_wgrad_any)
        elif ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dw, db = _wgrad_any(dy2, x2, y, ctx.relu_in, ctx.has_bias)
        if ctx.needs_input_grad[0]:
            link = ctx.grad_sum
            if ctx.t_ok and dy2.dtype is BF16:
                dx = run(dy2, wpt, K, None, link.buf if link is not None else None, y, x2 if ctx.relu_in else None,
                         False, False)
            else:
                # narrow output layer (fc_out: N = 3): dX through the fp32 kernels on casts, rounded to bf16 storage
                dyf = dy2.float()
                if y is not None:
                    dyf = dyf * (y > 0)
                dxf = hip_linear.linear(dyf, w_plain.t().contiguous())          # [M,N] x [N,K]
                if ctx.relu_in:
                    dxf = dxf * (x2 > 0)
                if link is not None and link.buf is not None:
                    dxf = dxf + link.buf.float()
                dx = dxf.to(BF16)
            dx = dx.reshape(ctx.x_shape)
            if link is not None:
                link.pending -= 1
                if link.pending > 0:
                    link.buf, dx = dx.reshape(-1, K), None
                else:
                    link.buf, link.pending = None, link.total
        if ctx.has_res and ctx.needs_input_grad[3]:
            dres = dy2 if y is None else dy2 * (y > 0)
            dres = dres.to(BF16).reshape(dy.shape)
        return dx, dw, db, dres, None, None, None, None, None, None, None


def k4_forward(x2, w4, b, relu_out):
    """bf16 Y [M,N] = act(x2 [M,4] fp32 @ w4 [N,4]^T + b)  (nsdp_linear_k4_bf16)."""
    M, N = x2.shape[0], w4.shape[0]
    y = torch.empty((M, N), dtype=BF16, device=x2.device)
    with on_device(x2):
        check(lib().nsdp_linear_k4_bf16(fptr(x2, "x"), fptr(w4, "weight"), optptr(b), ctypes.c_void_p(y.data_ptr()), _ll(M),
                                        _ci(N), _ci(int(relu_out)), stream_ptr()), "nsdp_linear_k4_bf16")
    return y


def k4_wgrad(dy2, x2, mask, relu_x, want_db):
    """dW [N,4], db [N] (fp32) from bf16 dY [M,N] and fp32 X [M,4] (nsdp_linear_wgrad_k4_bf16)."""
    assert not relu_x
    M, N = dy2.shape
    L = lib()
    L.nsdp_linear_wgrad_k4_bf16_workspace_bytes.restype = ctypes.c_size_t
    nbytes = int(L.nsdp_linear_wgrad_k4_bf16_workspace_bytes(_ll(M), _ci(N)))
    ws = torch.empty(max(nbytes // 4, 1), dtype=torch.float32, device=dy2.device)
    dw = torch.empty((N, 4), dtype=torch.float32, device=dy2.device)
    db = torch.empty((N,), dtype=torch.float32, device=dy2.device) if want_db else None
    with on_device(dy2):
        check(L.nsdp_linear_wgrad_k4_bf16(hptr(dy2, "dy"), fptr(x2, "x"), opthptr(mask, "mask"), fptr(dw), optptr(db), _ll(M),
                                          _ci(N), fptr(ws), ctypes.c_size_t(nbytes), stream_ptr()),
              "nsdp_linear_wgrad_k4_bf16")
    return dw, db


class _LinearK4B16Fn(torch.autograd.Function):
    """First layer of a position-encoding MLP with bf16 storage: fp32 coordinates [.., 3 or 4] in, bf16 [.., N] out."""

    @staticmethod
    def forward(ctx, x, w, b, relu_out, w_param, b_param):
        ctx.w_param, ctx.b_param = w_param, b_param
        Kx = x.shape[-1]
        K = w.shape[1]                         # the layer's own width: x may arrive zero-padded to 4 already (ops.relative_coords)
        if Kx != K and not (Kx == 4 and K == 3 and not ctx.needs_input_grad[0]):
            raise ValueError(f"linear_k4: input width {Kx} against a [{w.shape[0]}, {K}] weight")
        N = w.shape[0]
        x2 = x.reshape(-1, Kx)
        x2 = torch.nn.functional.pad(x2, (0, 4 - Kx)) if Kx < 4 else (x2 if x2.is_contiguous() else x2.contiguous())
        w4 = None
        if w_param is not None and K < 4:      # the zero-padded [N,4] weight is a cache of the parameter, like the packs
            key = hip_linear._pack_key(w_param)
            hit = w_param.__dict__.get("_nsdp_w4")
            if hit is not None and hit[0] == key:
                w4 = hit[1]
        if w4 is None:
            w4 = torch.nn.functional.pad(w, (0, 4 - K)) if K < 4 else w.contiguous()
            if w_param is not None and K < 4:
                w_param.__dict__["_nsdp_w4"] = (hip_linear._pack_key(w_param), w4)
        y = k4_forward(x2, w4, b, relu_out)
        ctx.k_orig, ctx.n_out, ctx.x_shape, ctx.has_bias = K, N, x.shape, b is not None
        ctx.save_for_backward(x2, y if relu_out else None, w4 if ctx.needs_input_grad[0] else None)
        return y.reshape(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, y, w4 = ctx.saved_tensors
        N, K = ctx.n_out, ctx.k_orig
        dy2 = dy.reshape(-1, N)
        dy2 = dy2 if dy2.is_contiguous() else dy2.contiguous()
        dx = dw = db = None

        def wg(dy2_, x2_, mask_, relu_x_, want_db_, out_=None):      # (padded K: never accumulates in place)
            gw, gb = k4_wgrad(dy2_, x2_, mask_, relu_x_, want_db_)
            return (gw[:, :K].contiguous() if K < 4 else gw), gb
        if ctx.w_param is not None:
            if hip_linear._use_side_stream(dy2):
                hip_linear._wgrad_deferred(dy2, x2, y, False, K, ctx.w_param, ctx.b_param, fn=wg)
            else:
                hip_linear.wgrad_direct(dy2, x2, y, False, K, ctx.w_param, ctx.b_param, fn=wg)
        elif ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dw, db = wg(dy2, x2, y, False, ctx.has_bias)
        if ctx.needs_input_grad[0]:
            # dX [M,4] = dY' [M,N] W [N,4]: an ordinary bf16 layer with 4 (fp32) outputs
            _, wpt = pack_weight_b16(w4, False, True)
            dx4 = run(dy2, wpt, 4, None, None, y, None, False, False, out_f32=True)
            dx = (dx4[:, :K] if K < 4 else dx4).reshape(ctx.x_shape)
        return dx, dw, db, None, None, None


def k4_supported(x, N, relu_in, residual):
    return x.dtype is torch.float32 and x.shape[-1] in (3, 4) and N % 8 == 0 and 8 <= N <= 256 and not relu_in and residual is None


def linear_k4(x, weight, bias, relu_out, w_param, b_param):
    w2 = weight.squeeze(-1) if weight.dim() == 3 else weight
    if w_param is not None:
        return _LinearK4B16Fn.apply(x, w2.detach(), None if bias is None else bias.detach(), bool(relu_out), w_param, b_param)
    return _LinearK4B16Fn.apply(x, w2, bias, bool(relu_out), None, None)


def linear(x, weight, bias, relu_in, relu_out, residual, w_param, b_param, grad_sum, owner, out_f32=False):
    w2 = weight.squeeze(-1) if weight.dim() == 3 else weight
    if w_param is not None:
        return _LinearB16Fn.apply(x, w2.detach(), None if bias is None else bias.detach(), residual, bool(relu_in),
                                  bool(relu_out), w_param, b_param, grad_sum, owner, bool(out_f32))
    return _LinearB16Fn.apply(x, w2, bias, residual, bool(relu_in), bool(relu_out), None, None, grad_sum, owner,
                              bool(out_f32))
