"""BatchNorm1d on channels-last rows with the fused residual-add prologue and ReLU epilogue
(csrc/batchnorm.hip).  Semantics of nn.BatchNorm1d (momentum update, biased variance for normalisation,
unbiased for running_var, eval mode on running statistics)."""
from __future__ import annotations

import contextlib
import os
import ctypes

import torch

from ._lib import check, fptr, lib, on_device, optptr, stream_ptr

_RUNNING_UPDATES = 1
_stats_epoch = 0        # advanced by every training-mode norm: keys the cached inference constants (batch_norm)


_frozen_capture = None      # a list while a GraphedStep(weights_change=False) captures: the cached constants its graph reads


class frozen_capture:
    """Context of a capture over FROZEN weights and buffers: the cached inference constants may be read (their addresses go into
    the graph); ``keep`` receives the tensors so that the graph's owner keeps them alive."""

    def __init__(self, keep):
        self.keep = keep

    def __enter__(self):
        global _frozen_capture
        self._was, _frozen_capture = _frozen_capture, self.keep
        return self

    def __exit__(self, *exc):
        global _frozen_capture
        _frozen_capture = self._was
        return False


def invalidate_inference_constants():
    """Declare every cached 1 / sqrt(running_var + eps) stale.  Training-mode norms do it themselves; a REPLAY of a captured
    train step runs them without any Python (nsdp_amd.graph_step calls this after every replay, next to the weight packs)."""
    global _stats_epoch
    _stats_epoch += 1


@contextlib.contextmanager
def running_updates(n: int):
    """Inside this context a training-mode norm folds its batch statistics into the running statistics ``n`` times and
    counts ``n`` batches: what ``n`` forward passes of the module over the SAME input leave behind.  FlowArbitrary's
    reference encodes one cloud twice per step (model/flow_arbitrary.py:19-20); this library encodes it once."""
    global _RUNNING_UPDATES
    prev, _RUNNING_UPDATES = _RUNNING_UPDATES, int(n)
    try:
        yield
    finally:
        _RUNNING_UPDATES = prev


NATIVE_BF16 = True        # bf16-storage kernels (False: ops.batch_norm casts around the fp32 ones -- reference semantics for tests)
BF16 = torch.bfloat16


def _p(t, dtype, name="tensor"):
    if t is None:
        return ctypes.c_void_p(0)
    if t.dtype is not dtype or not t.is_cuda or not t.is_contiguous():
        raise TypeError(f"{name}: expected a contiguous GPU {dtype} tensor, got {t.dtype}")
    return ctypes.c_void_p(t.data_ptr())


def _fn(name, dtype):
    return getattr(lib(), name + ("_bf16" if dtype is BF16 else ""))

_ll = ctypes.c_longlong
_ci = ctypes.c_int
_cf = ctypes.c_float


def _ws(C, device):
    L = lib()
    L.nsdp_bn_workspace_bytes.restype = ctypes.c_size_t
    return torch.empty(int(L.nsdp_bn_workspace_bytes(_ci(C))) // 4, dtype=torch.float32, device=device)


# Parameter gradients into EXISTING .grad buffers (the data-parallel flat-bucket views, gradient accumulation over
# micro-batches): autograd's AccumulateGrad would add each of the 70 scale / shift gradients of a step with a kernel of its own
# (a captured data-parallel step had 761 nodes against the plain step's 681, +0.5 ms at B = 32).  Here a norm whose two
# parameters already carry a .grad parks its gradients; ONE torch._foreach_add_ at the end of the backward pass adds them all.
# Same private engine hooks as hip_linear's side-stream publication; without them (or with observers on the parameters) the
# gradients go through autograd as ever.  NSDP_BN_DIRECT_GRADS=0: always through autograd (A/B knob).
_BN_DIRECT = os.environ.get("NSDP_BN_DIRECT_GRADS", "1") != "0"
_pending_grads = {}      # (device index, autograd graph task) -> ([.grad buffers], [gradients])


def _flush_param_grads(key):
    dst, src = _pending_grads.pop(key, ([], []))
    if dst:
        with torch.no_grad():
            torch._foreach_add_(dst, src)


def _park_param_grads(gamma, beta, dgamma, dbeta):
    """True when the pair was parked for the end-of-backward add (the caller then reports no gradient to autograd)."""
    from . import hip_linear
    if not (_BN_DIRECT and hip_linear._HAVE_ENGINE_HOOKS and hip_linear._PARAM_GRADS_DIRECT):
        return False
    if not (isinstance(gamma, torch.nn.Parameter) and isinstance(beta, torch.nn.Parameter) and gamma.grad is not None
            and beta.grad is not None and gamma.grad.dtype is torch.float32 and beta.grad.dtype is torch.float32
            and not hip_linear._observed(gamma) and not hip_linear._observed(beta)):
        return False
    key = (dgamma.device.index, hip_linear._graph_task_id())
    slot = _pending_grads.get(key)
    if slot is None:
        slot = _pending_grads[key] = ([], [])
        torch.autograd.Variable._execution_engine.queue_callback(lambda: _flush_param_grads(key))
    slot[0].extend((gamma.grad, beta.grad))
    slot[1].extend((dgamma, dbeta))
    return True


class _BatchNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, addend, gamma, beta, running_mean, running_var, training, momentum, eps, relu, nbt=None, updates=1,
                infer=None):
        ctx.params = (gamma, beta)
        shape = x.shape
        C = shape[-1]
        x2 = x.reshape(-1, C)
        x2 = x2 if x2.is_contiguous() else x2.contiguous()
        a2 = None
        if addend is not None:
            a2 = addend.reshape(-1, C)
            a2 = a2 if a2.is_contiguous() else a2.contiguous()
        R = x2.shape[0]
        dev = x2.device
        dt = x2.dtype
        if a2 is not None and a2.dtype is not dt:
            a2 = a2.to(dt)
        with on_device(x2):
            y = torch.empty_like(x2)
            if training:
                # statistics + running-average update + normalisation: one launch for R <= 16384 rows (csrc/batchnorm.hip)
                stats = torch.empty(2, C, dtype=torch.float32, device=dev)
                mean, invstd = stats[0], stats[1]
                check(_fn("nsdp_bn_train_fwd", dt)(_p(x2, dt, "x"), _p(a2, dt, "addend"), _ll(R), _ci(C), _cf(eps),
                                                   _cf(momentum), _ci(updates), optptr(running_mean), optptr(running_var),
                                                   optptr(nbt), fptr(gamma, "weight"), fptr(beta, "bias"), _ci(int(relu)),
                                                   _p(y, dt), fptr(mean), fptr(invstd), fptr(_ws(C, dev)), stream_ptr()),
                      "nsdp_bn_train_fwd")
            else:
                mean = running_mean
                invstd = infer if infer is not None else torch.rsqrt(running_var + eps)
                check(_fn("nsdp_bn_apply", dt)(_p(x2, dt), _p(a2, dt), fptr(mean), fptr(invstd), fptr(gamma, "weight"),
                                               fptr(beta, "bias"), _ll(R), _ci(C), _ci(int(relu)), _p(y, dt), stream_ptr()),
                      "nsdp_bn_apply")
        ctx.save_for_backward(x2, a2, gamma, mean, invstd, y if relu else None)
        ctx.training, ctx.shape, ctx.has_addend = bool(training), shape, addend is not None
        return y.reshape(shape)

    @staticmethod
    def backward(ctx, dy):
        x2, a2, gamma, mean, invstd, y = ctx.saved_tensors
        R, C = x2.shape
        dy2 = dy.reshape(R, C)
        dy2 = dy2 if dy2.is_contiguous() else dy2.contiguous()
        dev = dy2.device
        dt = x2.dtype
        if dy2.dtype is not dt:
            dy2 = dy2.to(dt)
        dx = torch.empty_like(x2)
        dgamma = torch.empty(C, dtype=torch.float32, device=dev)
        dbeta = torch.empty(C, dtype=torch.float32, device=dev)
        with on_device(dy2):
            check(_fn("nsdp_bn_backward", dt)(_p(dy2, dt, "dy"), _p(y, dt), _p(x2, dt), _p(a2, dt), fptr(mean), fptr(invstd),
                                              fptr(gamma), _ll(R), _ci(C), _ci(int(ctx.training)), _p(dx, dt), fptr(dgamma),
                                              fptr(dbeta), fptr(_ws(C, dev)), stream_ptr()), "nsdp_bn_backward")
        dx = dx.reshape(ctx.shape)
        if ctx.needs_input_grad[2] and ctx.needs_input_grad[3] and _park_param_grads(ctx.params[0], ctx.params[1], dgamma, dbeta):
            dgamma = dbeta = None
        return dx, (dx if ctx.has_addend else None), dgamma, dbeta, None, None, None, None, None, None, None, None, None


def batch_norm(x, bn: torch.nn.BatchNorm1d, addend=None, relu=False):
    """relu?( BN(x + addend) ) with bn's parameters / running statistics / mode."""
    training = bn.training or not bn.track_running_stats
    counts = bn.training and bn.track_running_stats
    updates = _RUNNING_UPDATES if counts else 1
    nbt = None
    if counts and bn.momentum is not None and bn.num_batches_tracked.is_cuda:
        nbt = bn.num_batches_tracked          # incremented by the norm's own kernel (35 tiny add kernels per step otherwise)
    elif counts:
        if updates != 1 and bn.momentum is None:
            raise NotImplementedError("running_updates(n > 1) with momentum=None (cumulative average)")
        bn.num_batches_tracked.add_(updates)
    rm = bn.running_mean if bn.track_running_stats else None
    rv = bn.running_var if bn.track_running_stats else None
    if bn.momentum is None:
        # nn.BatchNorm's cumulative moving average: factor 1 / num_batches_tracked (already counted above); this reads the
        # counter back from the device, i.e. one host sync per call -- none of the NSDP configurations uses it
        momentum = 1.0 / max(1.0, float(bn.num_batches_tracked)) if (bn.training and bn.track_running_stats) else 0.0
    else:
        momentum = float(bn.momentum)
    infer = None
    if training:
        global _stats_epoch
        _stats_epoch += 1          # (the kernels rewrite the running statistics behind the tensors' version counters)
    elif not torch.is_grad_enabled() and not torch.cuda.is_current_stream_capturing():
        # inference: 1 / sqrt(running_var + eps) is a constant of the module until its buffers change -- computed once (by the
        # same two torch kernels as ever: FlowArbitrary's second network turns a one-ulp difference in the first one's norms into
        # flipped neighbours), cached on the module, instead of an add + rsqrt launch in front of each of the 35 norms of every
        # forward.  (Inside a capture the cache is only READ: a tensor created there lives in the graph's pool.)
        key = (rv.data_ptr(), rv._version, _stats_epoch, float(bn.eps))
        hit = bn.__dict__.get("_nsdp_invstd")
        if hit is None or hit[0] != key:
            hit = bn.__dict__["_nsdp_invstd"] = (key, torch.rsqrt(rv + float(bn.eps)))
        infer = hit[1]
    elif not torch.is_grad_enabled() and _frozen_capture is not None:
        # inside a capture whose owner declared the weights FROZEN (GraphedStep(weights_change=False)): the cached constant's
        # address goes into the graph, which keeps the tensor alive (_frozen_capture collects it).  Any other capture -- an eval
        # step replayed while training continues -- computes the rsqrt as a node of the graph: a cached tensor would be frozen
        # into every replay next to the LIVE running_mean, and freed under the graph by the next eager evaluation.
        hit = bn.__dict__.get("_nsdp_invstd")
        if hit is not None and hit[0] == (rv.data_ptr(), rv._version, _stats_epoch, float(bn.eps)):
            infer = hit[1]
            _frozen_capture.append(infer)
    return _BatchNormFn.apply(x, addend, bn.weight, bn.bias, rm, rv, training, momentum, float(bn.eps), bool(relu), nbt,
                              updates, infer)
