"""HipAdam -- the optimizer step of the train step as ONE launch of csrc/adam.hip.

The reference builds `torch.optim.Adam(parameters, lr, weight_decay)` (/root/reference/model/__init__.py:10-41) and calls
`optimizer.step()` once per train step (model/deformation_networks.py:63-77, model/flow_arbitrary.py:30-48).  This class IS a
`torch.optim.Adam` as far as everything around the step is concerned -- the same `param_groups`, the same `state`
(`step`, `exp_avg`, `exp_avg_sq` per parameter), the same `state_dict()` / `load_state_dict()`, so optimizer checkpoints
travel both ways (train.py:118-131 of the reference saves `opt_*` files) -- with `step()` replaced: the update of every
parameter tensor that has a gradient runs as one `nsdp_adam_multi_f32` launch over tables in device memory.  The step
counters live on the device (torch's `capturable` layout), the learning rate may be a device tensor
(`graph_step.set_lr`): the step is capturable as it is.  Parameters must be fp32 tensors on one GPU; there is no CPU
path here (CPU parameters get `torch.optim.Adam` from `optimizer_factory`)."""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from ._lib import check, lib, stream_ptr


class _AdamDesc(ctypes.Structure):       # NsdpAdamDesc (include/nsdp_hip.h)
    _fields_ = [("param", ctypes.c_void_p), ("grad", ctypes.c_void_p), ("exp_avg", ctypes.c_void_p),
                ("exp_avg_sq", ctypes.c_void_p), ("step", ctypes.c_void_p), ("numel", ctypes.c_longlong)]


_DESC_DTYPE = np.dtype([("param", "<u8"), ("grad", "<u8"), ("exp_avg", "<u8"), ("exp_avg_sq", "<u8"), ("step", "<u8"),
                        ("numel", "<i8")])
assert _DESC_DTYPE.itemsize == ctypes.sizeof(_AdamDesc)


# While nsdp_amd.graph_step.GraphedStep captures a step this is ITS list: a pinned table buffer that a captured optimizer step
# copies from belongs to that graph and is released when the graph is closed (outside a GraphedStep capture -- plain
# torch.cuda.graph -- the buffers stay with the optimizer for its lifetime, HipAdam._graph_hosts).
_capture_hosts = None


class HipAdam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, maximize=False):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, maximize=maximize,
                         foreach=False, capturable=True)
        for group in self.param_groups:
            for p in group["params"]:
                if not (p.is_cuda and p.dtype == torch.float32):
                    raise ValueError("HipAdam: parameters must be fp32 tensors on a GPU (got %s on %s)" % (p.dtype, p.device))
        self._plans = {}
        self._spares = []             # pinned buffers for the tables of the next captured steps (see _plan)
        self._graph_hosts = []        # pinned buffers that captured graphs copy their tables from (alive as long as we are)

    # torch.optim.Adam.__setstate__ rewrites the step counters for its own kernels; nothing to add here.

    def _state_of(self, p):
        st = self.state[p]
        if len(st) == 0:
            st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        elif not (torch.is_tensor(st["step"]) and st["step"].is_cuda and st["step"].dtype == torch.float32):
            # (a state dict written by a non-capturable torch Adam and assigned by hand: counter on the host)
            st["step"] = torch.as_tensor(float(st["step"]), dtype=torch.float32, device=p.device)
        return st

    @staticmethod
    def _table_bytes_max(group, chunk):
        n = len(group["params"])
        return n * _DESC_DTYPE.itemsize + 8 * sum((p.numel() + chunk - 1) // chunk for p in group["params"])

    _N_SPARES = 8       # captures that may follow one another with no eager step in between (25 KB of pinned memory each)

    def _refill_spares(self):
        if len(self._spares) >= self._N_SPARES:
            return
        chunk = int(lib().nsdp_adam_chunk_elems())
        nbytes = max(self._table_bytes_max(g, chunk) for g in self.param_groups)
        while len(self._spares) < self._N_SPARES:
            self._spares.append(torch.empty(nbytes, dtype=torch.uint8).pin_memory())

    def _plan(self, gi, group):
        """Device tables of one parameter group, rebuilt only when a tensor of it moved (gradients set to None and
        re-created by the caching allocator normally come back at the same addresses; under graph replay nothing moves)."""
        rows, keep = [], []
        for p in group["params"]:
            g = p.grad
            if g is None:
                continue
            if g.is_sparse:
                raise RuntimeError("HipAdam does not support sparse gradients")
            st = self._state_of(p)
            m, v = st["exp_avg"], st["exp_avg_sq"]
            if not (p.is_contiguous() and g.is_contiguous() and m.is_contiguous() and v.is_contiguous()):
                raise RuntimeError("HipAdam: parameters, gradients and moments must be contiguous")
            if g.dtype != torch.float32 or g.device != p.device:
                raise RuntimeError("HipAdam: gradient of dtype %s on %s for an fp32 parameter on %s" % (g.dtype, g.device, p.device))
            if p.numel() == 0:
                continue
            rows.append((p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), st["step"].data_ptr(), p.numel()))
            keep.append(st["step"])
        if not rows:
            return None
        key = tuple(rows)
        plan = self._plans.get(gi)
        if plan is not None and plan["key"] == key:
            return plan
        dev = group["params"][0].device
        chunk = int(lib().nsdp_adam_chunk_elems())
        descs = np.array(rows, dtype=_DESC_DTYPE)
        counts = (descs["numel"] + chunk - 1) // chunk
        tensor_of = np.repeat(np.arange(len(rows), dtype=np.int32), counts)
        first = np.cumsum(counts) - counts
        chunk_of = (np.arange(tensor_of.size, dtype=np.int64) - np.repeat(first, counts)).astype(np.int32)
        # one tensor after the other: neighbours in the grid stream neighbouring addresses
        chunks = np.stack([tensor_of, chunk_of], axis=1).astype(np.int32)
        blob = np.concatenate([descs.view(np.uint8).reshape(-1), chunks.view(np.uint8).reshape(-1)])
        assert descs.nbytes % 8 == 0
        if torch.cuda.is_current_stream_capturing():
            # The gradients of a captured step live in the graph's pool: the tables are rebuilt INSIDE the capture.  Their
            # upload becomes a copy node from a pinned buffer that belongs to this graph from now on (allocated ahead of
            # the capture: pinned allocations are not allowed while capturing), the counters' fill a memset node.
            host = self._spares.pop() if self._spares else None
            if host is None or host.numel() < blob.size:
                host = torch.empty(blob.size, dtype=torch.uint8).pin_memory()      # (fails loudly if the runtime refuses)
            host[:blob.size] = torch.from_numpy(blob)
            # Invariant the captured graph relies on: EVERY replay re-uploads the table from this buffer (a copy node) and
            # re-zeroes `done` (a memset node), so an eager step that later overwrites self._plans[gi] cannot disturb it.
            (_capture_hosts if _capture_hosts is not None else self._graph_hosts).append(host)
            table = torch.empty(blob.size, dtype=torch.uint8, device=dev)
            table.copy_(host[:blob.size], non_blocking=True)
        else:
            table = torch.from_numpy(blob).to(dev)
        done = torch.zeros(len(rows), dtype=torch.int32, device=dev)
        plan = {"key": key, "table": table, "done": done, "n_chunks": int(chunks.shape[0]),
                "descs_ptr": table.data_ptr(), "chunks_ptr": table.data_ptr() + descs.nbytes, "steps": keep}
        self._plans[gi] = plan
        return plan

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if not torch.cuda.is_current_stream_capturing():
            self._refill_spares()
        for gi, group in enumerate(self.param_groups):
            if group.get("amsgrad", False):
                raise RuntimeError("HipAdam: amsgrad is not offered (the reference does not use it)")
            plan = self._plan(gi, group)
            if plan is None:
                continue
            lr = group["lr"]
            lr_dev, lr_host = (lr.data_ptr(), 0.0) if torch.is_tensor(lr) and lr.is_cuda else (None, float(lr))
            if lr_dev is not None and lr.dtype != torch.float32:
                raise RuntimeError("HipAdam: a tensor learning rate must be fp32")
            beta1, beta2 = group["betas"]
            dev = group["params"][0].device
            with torch.cuda.device(dev):
                check(lib().nsdp_adam_multi_f32(
                    ctypes.c_void_p(plan["descs_ptr"]), ctypes.c_void_p(plan["chunks_ptr"]), ctypes.c_int(plan["n_chunks"]),
                    ctypes.c_void_p(plan["done"].data_ptr()), ctypes.c_void_p(lr_dev), ctypes.c_double(lr_host),
                    ctypes.c_double(float(beta1)), ctypes.c_double(float(beta2)), ctypes.c_double(float(group["eps"])),
                    ctypes.c_double(float(group["weight_decay"])), ctypes.c_int(1 if group.get("maximize", False) else 0),
                    stream_ptr()), "nsdp_adam_multi_f32")
            # written behind autograd's back: saved-tensor checks of a retained graph must see the parameters change
            torch.autograd.graph.increment_version([p for p in group["params"] if p.grad is not None])
        return loss
