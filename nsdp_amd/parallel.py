"""Data parallelism for the TDNet train step: one process per GPU, shapes sharded across ranks, ONE
exchange per step -- a sum all-reduce of the flat fp32 gradient over RCCL/xGMI.

The reference is single-process (no DDP, no SyncBN): BatchNorm statistics stay per-rank here too.
Every parameter's ``.grad`` is a view into one flat buffer (decoder parameters first: their gradients
are complete before the encoder's backward starts), so the collective runs in place with no
flatten/unflatten copies: 17.97 MB for the forward model, 35.94 MB for FlowArbitrary.  Parameters that
receive no gradient (``transformer_begin.w_qs/w_ks/w_vs`` of the 'backward' net) simply keep a zero
slice -- the same on every rank -- instead of needing DDP's ``find_unused_parameters``.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class GradAllReducer:
    def __init__(self, model: torch.nn.Module, world_size: int, process_group=None, first=("decoder",),
                 always_exchange: bool = False):
        """``always_exchange``: run the collectives even in a communicator of one rank (the sum over one rank is the
        identity; used to exercise the RCCL path -- communicator, in-place all-reduce on the flat bucket's two halves --
        on a single GPU, tests/test_rccl_gpu.py and ``bench.py --force-reducer``)."""
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        head = [(n, p) for n, p in named if any(("." + f + ".") in ("." + n) for f in first)]
        tail = [(n, p) for n, p in named if not any(("." + f + ".") in ("." + n) for f in first)]
        self.named = head + tail
        self.world_size = int(world_size)
        self.group = process_group
        self.always_exchange = bool(always_exchange)
        total = sum(p.numel() for _, p in self.named)
        ref = self.named[0][1]
        self.flat = torch.zeros(total, dtype=ref.dtype, device=ref.device)
        self.views = []
        off = 0
        for _, p in self.named:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        self.split = sum(p.numel() for _, p in head)  # bucket boundary: [0, split) = decoder
        self.zero_grad()

    @property
    def nbytes(self) -> int:
        return self.flat.numel() * self.flat.element_size()

    def zero_grad(self):
        """Replaces optimizer.zero_grad(): zero the flat buffer and (re-)attach the views as .grad."""
        self.flat.zero_()
        for (_, p), v in zip(self.named, self.views):
            p.grad = v

    def adopt_grads(self):
        """Make every ``.grad`` a view into the flat buffer again.  ``optimizer.zero_grad()`` (default
        ``set_to_none=True``; the shipped ``train_on_batch_*`` functions call it) detaches the views: backward then
        creates fresh gradient tensors and the flat buffer still holds the previous step.  A missing gradient becomes a
        zero slice, a stray one is copied into its slice and replaced by the view.  Returns the number of parameters
        that had to be re-attached (0 on the fast path: ``reducer.zero_grad()`` instead of ``optimizer.zero_grad()``)."""
        n = 0
        for (_, p), v in zip(self.named, self.views):
            g = p.grad
            if g is None:
                v.zero_()
            elif g.data_ptr() != v.data_ptr() or g.shape != v.shape:
                v.copy_(g)
            else:
                continue
            p.grad = v
            n += 1
        return n

    def all_reduce_mean(self):
        """Sum over ranks, then divide by world size (mean of per-rank mean losses = the global mean
        loss when every rank holds the same number of shapes).  Runs after ``backward()`` has returned -- the
        weight gradients are published at the end of the backward pass (hip_linear side stream), so there is nothing
        to overlap with; the two buckets only pipeline with each other (18 MB: ~0.2 ms over xGMI)."""
        self.adopt_grads()
        if self.world_size > 1 or (self.always_exchange and dist.is_available() and dist.is_initialized()):
            if self.split and self.split < self.flat.numel():
                h1 = dist.all_reduce(self.flat[:self.split], group=self.group, async_op=True)
                h2 = dist.all_reduce(self.flat[self.split:], group=self.group, async_op=True)
                h1.wait()
                h2.wait()
            else:
                dist.all_reduce(self.flat, group=self.group)
            if self.world_size > 1:
                self.flat.div_(self.world_size)


def data_parallel_step(train_on_batch, reducer: GradAllReducer):
    """Wrap one of the reference-shaped step functions (``train_on_batch(model, optimizer, data_dict, config)``:
    zero_grad, forward, loss, backward, optimizer.step) for data parallelism WITHOUT changing it: the gradient exchange
    is run by a pre-hook of ``optimizer.step`` -- i.e. after backward, before the update -- and the flat views are
    re-attached after the function's own ``optimizer.zero_grad()`` detached them.  Usage:

        step = data_parallel_step(train_on_batch, GradAllReducer(model, world))
        loss = step(model, optimizer, data_dict, config)      # local loss of this rank
    """
    def step(model, optimizer, data_dict, config):
        handle = optimizer.register_step_pre_hook(lambda *_: reducer.all_reduce_mean())
        try:
            return train_on_batch(model, optimizer, data_dict, config)
        finally:
            handle.remove()
    return step


class DataParallel:
    """The data-parallel side of the train harness (nsdp_amd.train.fit), one instance per rank.  Policy, stated once:

    * shapes are sharded by BATCH: of every ``world`` consecutive batches of a loader, rank r takes the r-th (the global
      batch is the reference's batch x world); a trailing group of fewer than ``world`` batches is dropped, so every rank
      takes the same number of steps (the gradient exchange is a collective);
    * one exchange per step: the mean of the flat fp32 gradient (GradAllReducer), then the identical Adam update on every
      rank -- weights stay bit-equal across ranks (``in_sync``);
    * BatchNorm: batch statistics are per rank (the reference has no SyncBN).  The running buffers are therefore per rank
      too; rank 0's are THE model's (as with torch DDP's ``broadcast_buffers``): they are broadcast to every rank before a
      validation pass and before a checkpoint, so every rank evaluates -- and rank 0 saves -- the same model;
    * files are written by rank 0 only; every rank reads them at resume (same node), and the initial weights are broadcast
      from rank 0 on top, so the ranks start identical whatever they loaded or initialised."""

    def __init__(self, model: torch.nn.Module, rank: int, world_size: int, process_group=None, always_exchange: bool = False):
        self.rank, self.world_size, self.group = int(rank), int(world_size), process_group
        self.reducer = GradAllReducer(model, world_size, process_group, always_exchange=always_exchange)

    @property
    def is_main(self) -> bool:
        return self.rank == 0

    def _active(self) -> bool:
        return self.world_size > 1 and dist.is_available() and dist.is_initialized()

    def _broadcast(self, tensors):
        """rank 0's values into every rank's tensors, coalesced per dtype (one collective per dtype instead of ~400)."""
        if not self._active():
            return
        by_dtype = {}
        for t in tensors:
            by_dtype.setdefault(t.dtype, []).append(t)
        with torch.no_grad():
            for ts in by_dtype.values():
                flat = torch.cat([t.detach().reshape(-1) for t in ts])
                dist.broadcast(flat, src=0, group=self.group)
                off = 0
                for t in ts:
                    t.copy_(flat[off:off + t.numel()].view_as(t))
                    off += t.numel()

    def broadcast_model(self, model: torch.nn.Module):
        """Parameters AND buffers from rank 0 (start of training / after a resume)."""
        self._broadcast(list(model.parameters()) + list(model.buffers()))
        self._weights_rewritten()

    def broadcast_buffers(self, model: torch.nn.Module):
        """BatchNorm running statistics and counters from rank 0 (before validation / a checkpoint)."""
        self._broadcast(list(model.buffers()))

    @staticmethod
    def _weights_rewritten():
        from . import hip_linear
        hip_linear.invalidate_weight_packs()      # (parameters changed behind the optimizer's back)

    def shard(self, loader):
        """This rank's batches of one pass over ``loader`` (see the class docstring), lazily: one batch is held at a time.

        A loader that can produce a rank's share ITSELF -- a ``shard(rank, world_size)`` method returning an iterable, e.g. on
        top of a ``DistributedSampler`` with a per-epoch seed -- is asked to: nothing is then loaded and thrown away.
        Otherwise every rank walks the whole loader and keeps every world-th batch, which REQUIRES that the loader yields the
        same sequence on every rank (a shuffling loader must be seeded identically on all ranks: ranks that disagree on the order
        would train on overlapping / missing samples without any error) and costs world x the I/O."""
        own = getattr(loader, "shard", None)
        if callable(own):
            return own(self.rank, self.world_size)
        return self._every_nth(loader)

    def _every_nth(self, loader):
        mine = None
        for i, batch in enumerate(loader):
            pos = i % self.world_size
            if pos == self.rank:
                mine = batch
            if pos == self.world_size - 1:      # the group is complete on every rank: all of them step (a trailing partial
                yield mine                      # group is dropped -- the gradient exchange is a collective)
                mine = None

    def mean(self, value: float, device="cpu") -> float:
        """Mean over ranks of a per-rank scalar (epoch losses, validation losses)."""
        if not self._active():
            return float(value)
        t = torch.tensor([float(value)], dtype=torch.float64, device=device)
        dist.all_reduce(t, group=self.group)
        return float(t.item()) / self.world_size

    def in_sync(self, model: torch.nn.Module) -> bool:
        """Do all ranks hold bit-identical parameters?  (min == max over ranks of every element)"""
        if not self._active():
            return True
        w = torch.cat([p.detach().reshape(-1).float() for p in model.parameters()])
        lo, hi = w.clone(), w.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
        return bool(torch.equal(lo, hi))

    def wrap(self, train_on_batch):
        """The reference-shaped step function with the gradient exchange in front of ``optimizer.step``."""
        return data_parallel_step(train_on_batch, self.reducer)


def rank_device(local_rank: int, world_size: int, backend: str = "nccl"):
    """(device index, cpu mask) of this rank of a one-process-per-GPU job, the policy bench.py and nsdp_amd.train share:
    rank r -> GPU r; the ONE visible device when the launcher narrowed HIP_VISIBLE_DEVICES per process; with a non-RCCL
    backend (gloo: the path exercised on a single GPU) ranks wrap around the visible devices.  RCCL needs a device per rank.
    The process is pinned to its slice of the container's CPUs, taken from the GPU's NUMA node (cpu_budget.pin_rank)."""
    import os
    n_dev = torch.cuda.device_count()
    vis = [d for d in os.environ.get("HIP_VISIBLE_DEVICES", os.environ.get("CUDA_VISIBLE_DEVICES", "")).split(",") if d]
    if n_dev == 1 and world_size > 1 and backend == "nccl" and len(vis) == 1:
        index = 0
    elif backend != "nccl":
        index = local_rank % max(1, n_dev)
    else:
        if local_rank >= n_dev:
            raise RuntimeError(f"rank with LOCAL_RANK {local_rank} needs its own GPU for RCCL, {n_dev} visible "
                               "(--backend gloo shares devices between ranks)")
        index = local_rank
    torch.cuda.set_device(index)
    mask = None
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world_size)))
    if local_world > 1:
        from .cpu_budget import pin_rank
        props = torch.cuda.get_device_properties(index)
        bdf = (f"{getattr(props, 'pci_domain_id', 0):04x}:{props.pci_bus_id:02x}:{props.pci_device_id:02x}.0"
               if hasattr(props, "pci_bus_id") else None)
        mask = pin_rank(local_rank, local_world, bdf)
    return index, mask
