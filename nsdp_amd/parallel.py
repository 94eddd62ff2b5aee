"""Data parallelism for the TDNet train step: one process per GPU, shapes sharded across ranks, ONE
exchange per step -- the mean of the flat fp32 gradient over RCCL/xGMI, in two
buckets, the first of which travels UNDER the encoder's backward pass.

The reference is single-process (no DDP, no SyncBN): BatchNorm statistics stay per-rank here too.
Every parameter's ``.grad`` is a view into one flat buffer, the parameters of the LAST decoder of the
forward pass first (``decoder.*``; ``model_deform.decoder.*`` of FlowArbitrary): their gradients are
complete when that decoder's backward has run, i.e. before the encoder's backward starts.  The collective
runs in place on the two halves of the flat buffer with no flatten / unflatten copies: 17.97 MB for the
forward model, 35.94 MB for FlowArbitrary.  Parameters that receive no gradient
(``transformer_begin.w_qs/w_ks/w_vs`` of the 'backward' net) simply keep a zero slice -- the same on
every rank -- instead of needing DDP's ``find_unused_parameters``.

Overlap without hooks inside the backward pass: the autograd graph is CUT at the inputs of that decoder
(a forward pre-hook hands it detached leaves in place of the encoder's outputs and the query points, armed
by ``GradAllReducer.zero_grad(two_pass=True)`` for the next forward pass only) and the backward is run in two autograd
passes (``GradAllReducer.backward``) --

    head   loss.backward() down to the cut     -> the decoder's weight gradients are published (hip_linear's
                                                 end-of-backward callback of THIS pass), bucket 0 complete
           all-reduce(bucket 0), asynchronous  (torch.distributed's communication stream)
    tail   backward from the cut, with the gradients the leaves received
                                              -> encoder (and, in FlowArbitrary, the first network), bucket 1 complete
           all-reduce(bucket 1); wait for both; optimizer.step

-- which is also the shape a captured step takes: one graph per side of each collective (head / tail /
update; a collective cannot be captured), see bench.py and graph_step.GraphedTrainOnBatch.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def _cut(obj, pairs):
    """``obj`` with every tensor that requires a gradient replaced by a detached leaf that requires one (same storage: nothing
    is copied); the (original, leaf) pairs are appended to ``pairs``.  Tensors, dicts, lists and tuples, recursively."""
    if torch.is_tensor(obj):
        if obj.requires_grad and obj.is_floating_point():
            for orig, leaf in pairs:          # (one tensor passed twice: one leaf)
                if orig is obj:
                    return leaf
            leaf = obj.detach().requires_grad_(True)
            pairs.append((obj, leaf))
            return leaf
        return obj
    if isinstance(obj, dict):
        return {k: _cut(v, pairs) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_cut(v, pairs) for v in obj)
    return obj


class GradAllReducer:
    def __init__(self, model: torch.nn.Module, world_size: int, process_group=None, first=None,
                 always_exchange: bool = False):
        """``first``: name prefixes of the parameters of bucket 0 (default: the last decoder of the forward pass).
        ``always_exchange``: run the collectives even in a communicator of one rank (the mean over one rank is the
        identity; used to exercise the RCCL path -- communicator, in-place all-reduce on the flat bucket's two halves --
        on a single GPU, tests/test_rccl_gpu.py and ``bench.py --force-reducer``)."""
        if first is None:
            first = ("model_deform.decoder",) if hasattr(model, "model_deform") else ("decoder",)
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
        self.split = sum(p.numel() for _, p in head)  # bucket boundary: [0, split) = the early bucket
        # the cut of the two-pass backward: every tensor that requires a gradient among the inputs of the module whose
        # parameters are bucket 0, recorded by a forward pre-hook each time it runs
        self._armed = False          # cut the NEXT forward pass of the early-bucket module (zero_grad arms, the hook disarms)
        self._cut_pairs = None       # (original, leaf) of the pass that was cut, until backward_head took their gradients
        self._root_grads = None
        self._pending = [None, None]
        self._started = [False, False]
        self.enqueued_before_backward_returned = 0      # (tests: bucket-0 collectives issued between the two passes)
        mods = dict(model.named_modules())
        early = mods.get(first[0]) if len(first) == 1 else None
        self._can_cut = early is not None and bool(self.split) and self.split < total
        if self._can_cut:
            early.register_forward_pre_hook(self._cut_inputs, with_kwargs=True)
        self.zero_grad()

    def _cut_inputs(self, module, args, kwargs):
        if not self._armed or not torch.is_grad_enabled():
            return None
        self._armed = False
        pairs = []
        args, kwargs = _cut(tuple(args), pairs), _cut(dict(kwargs), pairs)
        if not pairs:
            return None
        self._cut_pairs = pairs
        return args, kwargs

    @property
    def nbytes(self) -> int:
        return self.flat.numel() * self.flat.element_size()

    def zero_grad(self, two_pass: bool = False):
        """Replaces optimizer.zero_grad(): zero the flat buffer and (re-)attach the views as .grad.  ``two_pass=True``:
        the step that starts here is finished with ``reducer.backward(loss)`` (or backward_head / backward_tail) -- the next
        forward pass of the early-bucket module is cut from what feeds it.  A step that calls ``loss.backward()`` itself
        (data_parallel_step) leaves it False: behind a cut, one backward pass stops at the decoder's inputs (``finish()``
        raises if a cut pass was never completed)."""
        self.flat.zero_()
        for (_, p), v in zip(self.named, self.views):
            p.grad = v
        self._armed = bool(two_pass) and self._can_cut
        self._cut_pairs = self._root_grads = None

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

    # ---- the exchange --------------------------------------------------------------------------------------------
    def exchanging(self) -> bool:
        return self.world_size > 1 or (self.always_exchange and dist.is_available() and dist.is_initialized())

    def _bucket(self, i):
        two = self.split and self.split < self.flat.numel()
        if not two:
            return self.flat if i == 1 else None      # one bucket: everything travels as "bucket 1", after the whole backward
        return self.flat[:self.split] if i == 0 else self.flat[self.split:]

    def _averaging(self) -> bool:
        # ncclAvg: the mean inside the collective (no division pass over the flat buffer afterwards); gloo has no AVG
        try:
            return dist.get_backend(self.group) == "nccl"
        except Exception:
            return False

    def start(self, i: int):
        """Issue bucket i's all-reduce now, asynchronously: torch.distributed runs it on the backend's own stream behind
        everything enqueued on the CURRENT stream so far, and the current stream goes on (with the encoder's backward, for
        bucket 0).  ``finish()`` makes the current stream wait for it."""
        if self._started[i]:
            return
        self._started[i] = True
        t = self._bucket(i)
        if t is None or not self.exchanging():
            return
        op = dist.ReduceOp.AVG if self._averaging() else dist.ReduceOp.SUM
        self._pending[i] = (dist.all_reduce(t, op=op, group=self.group, async_op=True), t, op)

    def finish(self):
        """Wait for whatever was started, start (and wait for) what was not; scale where the collective only summed.  After
        this the flat buffer holds the mean gradient on the current stream: ``optimizer.step()`` may follow."""
        if self._cut_pairs is not None or self._root_grads is not None:
            raise RuntimeError("GradAllReducer: the forward pass was cut at the decoder's inputs (zero_grad(two_pass=True)) but "
                               "the backward pass upstream of the cut never ran -- use reducer.backward(loss), or "
                               "backward_head(loss) + backward_tail()")
        for i in (0, 1):
            self.start(i)
        for i in (0, 1):
            pend, self._pending[i] = self._pending[i], None
            if pend is not None:
                work, t, op = pend
                work.wait()
                if op == dist.ReduceOp.SUM and self.world_size > 1:
                    t.div_(self.world_size)
        self._started = [False, False]

    def all_reduce_mean(self):
        """The whole exchange at once (after a one-pass ``backward()`` has returned): both buckets, then the wait.  Mean of
        per-rank mean losses = the global mean loss when every rank holds the same number of shapes."""
        self.adopt_grads()
        self.finish()

    # ---- the two-pass backward -----------------------------------------------------------------------------------
    def backward_head(self, loss):
        """First pass: ``loss.backward()`` down to the cut -- the early-bucket module's weight gradients are complete (and
        published) when it returns.  False when the forward pass was not cut (the step was not armed, the module did not run,
        nothing upstream needs a gradient): the caller runs the one-pass ``loss.backward()``."""
        self._armed = False
        pairs, self._cut_pairs = self._cut_pairs, None
        if not pairs:
            self._root_grads = None
            return False
        loss.backward()
        self._root_grads = [(orig, leaf.grad) for orig, leaf in pairs if leaf.grad is not None]
        return True

    def backward_tail(self):
        """Second pass: everything upstream of the cut."""
        pairs, self._root_grads = self._root_grads, None
        if pairs:
            # (the second pass decides like the first -- weight gradients on the side stream, compute units left free -- as ONE
            # backward pass would: hip_linear.inherit_pass_decisions)
            try:
                from . import hip_linear
                ctx = hip_linear.inherit_pass_decisions() if pairs[0][0].is_cuda else None
            except Exception:      # (CPU-only use of the reducer, no native library)
                ctx = None
            if ctx is None:
                torch.autograd.backward([r for r, _ in pairs], [g for _, g in pairs])
            else:
                with ctx:
                    torch.autograd.backward([r for r, _ in pairs], [g for _, g in pairs])

    def backward(self, loss):
        """``loss.backward()`` with bucket 0's all-reduce issued between the decoder's and the encoder's backward passes and
        bucket 1's right behind the second; ``finish()`` (or ``all_reduce_mean()``) completes the exchange."""
        if self.backward_head(loss):
            self.start(0)
            if self._pending[0] is not None:
                self.enqueued_before_backward_returned += 1
            self.backward_tail()
        else:
            loss.backward()
        self.start(1)


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

    @staticmethod
    def _fingerprint(batch) -> float:
        """A cheap order-sensitive checksum of a batch (dict / sequence of tensors)."""
        ts = list(batch.values()) if isinstance(batch, dict) else (list(batch) if isinstance(batch, (list, tuple)) else [batch])
        acc = 0.0
        for j, t in enumerate(t for t in ts if torch.is_tensor(t)):
            f = t.detach().reshape(-1)[:4096].double().cpu()
            acc += float((f * torch.arange(1, f.numel() + 1, dtype=torch.float64)).sum()) * (j + 1)
        return acc

    def _check_same_sequence(self, batch):
        """Every rank walks the whole loader in this fallback: they must see the SAME sequence (identically seeded shuffling),
        or they would train on overlapping / missing samples without any error.  Checked on the first batch of every pass."""
        if not self._active():
            return
        h = torch.tensor([self._fingerprint(batch)], dtype=torch.float64)
        lo, hi = h.clone(), h.clone()
        if dist.get_backend(self.group) == "nccl":
            lo, hi = lo.cuda(), hi.cuda()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
        if float(lo) != float(hi):
            raise RuntimeError("DataParallel.shard: the ranks' loaders do not yield the same sequence (first batch differs) -- seed "
                               "the shuffling identically on every rank, or give the loader a shard(rank, world_size) method")

    def _every_nth(self, loader):
        mine = None
        for i, batch in enumerate(loader):
            if i == 0:
                self._check_same_sequence(batch)
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
    # one visible device and several local ranks under RCCL: the launcher narrowed the devices per process -- by
    # HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES or, as is common on ROCm, ROCR_VISIBLE_DEVICES -- and that device is this rank's
    # (two ranks that really share one GPU are refused by RCCL itself when the communicator is built)
    vis = [d for name in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES")
           for d in os.environ.get(name, "").split(",") if d]
    if n_dev == 1 and world_size > 1 and backend == "nccl" and (vis or int(os.environ.get("LOCAL_WORLD_SIZE", "1")) > 1):
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
