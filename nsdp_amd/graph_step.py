"""A train / inference step captured once and replayed from C with the eager schedule's stream concurrency.

    step = GraphedStep(fn)            # fn(): enqueues one step on the current stream, returns tensors (e.g. the loss)
    step.capture(warmup=3)            # runs fn eagerly `warmup` times, then captures it (torch.cuda.graph: private pool)
    out = step()                      # one C call: ~1000 launches in ~4 ms of host time instead of 17-24 ms of Python

The reference enqueues its step op by op from Python (train.py:150-225); so does this repo's eager path, at 17-24 ms of
host time per forward.yaml step whatever the batch -- more than the GPU needs for a bf16 or small-batch step.  A plain
hipGraph replay (`torch.cuda.CUDAGraph.replay`) is no answer on this ROCm: it serialises the graph's branches (the
weight-gradient side stream, the geometry pyramid) and ends up SLOWER than eager (47.7 against 44.4 ms at B = 32).  Here the
captured hipGraph_t is handed to csrc/graph_exec.hip, which replays the same nodes on real HIP streams (chain
decomposition of the dependency graph, events on the cross-stream edges).

Rules of the capture (the same as for any CUDA / HIP graph):
  * shapes and addresses are frozen: `fn` must read its inputs from tensors that exist before `capture()` (write new
    batches into them with `copy_`), and what it returns are static tensors overwritten by every replay;
  * no host synchronisation inside `fn` (`.item()`, `.cpu()`, prints of tensors);
  * optimizers must be capturable (`capturable_adam`): the step counter lives on the device; a learning rate that changes
    must be a device tensor (`set_lr`); and the optimizer STATE must exist before the capture (run at least one eager
    step, `warmup >= 1`): state created inside a capture is re-initialised by every replay.
"""
from __future__ import annotations

import contextlib
import ctypes
import gc
import os

import torch

from ._lib import NsdpHipError, check, lib, stream_ptr


def capturable_adam(optimizer: torch.optim.Optimizer, lr_as_tensor: bool = True, fused: bool = True):
    """Switch a torch.optim.Adam(-like) optimizer to its capturable form IN PLACE (device-side step counter; the learning
    rate as a device tensor so that schedulers can change it between replays).  Call before the first step.
    ``fused``: PyTorch's fused multi-tensor Adam -- the capturable foreach form computes its bias corrections with ~15
    tensor-list operations, i.e. ~480 tiny kernels per step for the 350 parameter tensors of a TDNet (a replay of 1495 nodes
    instead of ~1020); the fused form is a dozen launches."""
    from .hip_adam import HipAdam
    if isinstance(optimizer, HipAdam):      # one launch of csrc/adam.hip, capturable as it is: only the learning rate moves
        fused = False
    for group in optimizer.param_groups:
        group["capturable"] = True
        if fused and "fused" in group and all(p.is_cuda and torch.is_floating_point(p) for p in group["params"]):
            group["fused"], group["foreach"] = True, False
        if lr_as_tensor and not torch.is_tensor(group["lr"]):
            dev = group["params"][0].device
            group["lr"] = torch.tensor(float(group["lr"]), dtype=torch.float32, device=dev)
    return optimizer


def set_lr(optimizer: torch.optim.Optimizer, value: float):
    """Change the learning rate of a `capturable_adam` optimizer between replays (fills the device tensor)."""
    for group in optimizer.param_groups:
        if torch.is_tensor(group["lr"]):
            group["lr"].fill_(float(value))
        else:
            group["lr"] = float(value)


_graveyard = []       # (executor handle, torch graph) of closed steps that could not be destroyed yet


def _bury():
    if not _graveyard or (torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()):
        return
    while _graveyard:
        handle, graph, pinned = _graveyard.pop()
        if handle:
            lib().nsdp_graph_exec_destroy(handle)
        del graph, pinned      # (the graph first: its copy nodes read the pinned buffers)


class GraphedStep:
    def __init__(self, fn, max_streams: int | None = None, weights_change: bool = True):
        """``weights_change`` (default True: a train step, or any step between whose replays the parameters may be updated):
        the weight-pack caches are declared stale before the capture -- their rebuild becomes a node of the graph -- and
        after every replay.  False is for inference over FROZEN weights only: the packs current at capture time are used
        by every replay (no repack per replay: 0.8 ms of a 5 ms eval step); whoever changes the weights must capture anew."""
        self.fn = fn
        self.weights_change = bool(weights_change)
        # (NSDP_GRAPH_STREAMS: A/B knob; 1 = everything on the caller's stream.  Measured at B = 32: 1 stream 46.5 ms = the sum
        # of the isolated kernel times, 2 streams 43.6, 3-6 streams 43.9; bf16 27.5 / 25.1 / 25.4)
        self.max_streams = int(max_streams if max_streams is not None else os.environ.get("NSDP_GRAPH_STREAMS", "2"))
        self._graph = None
        self._handle = ctypes.c_void_p(0)
        self._out = None
        self.info = None

    def capture(self, warmup: int = 3):
        if self._graph is not None:
            raise RuntimeError("already captured")
        if not hasattr(torch.cuda.CUDAGraph, "raw_cuda_graph"):
            raise NsdpHipError("this PyTorch cannot hand out the captured hipGraph_t (CUDAGraph.raw_cuda_graph)")
        # eager warm-up on a side stream (allocator steady state, weight packs, autotuned choices), as torch recommends
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(0, warmup)):
                self.fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        # Every host-side cache that mirrors the WEIGHTS (the fragment-major packs of hip_linear / hip_linear_bf16, the
        # fused decoder's pack) is declared stale here, so that their rebuild becomes a node at the head of the captured
        # step: a replay runs no Python -- a pack that happened to be valid at capture time (an eval forward just before)
        # would otherwise be frozen into every replay.
        from . import hip_linear
        if self.weights_change:
            hip_linear.invalidate_weight_packs()
        graph = torch.cuda.CUDAGraph(keep_graph=True)
        # thread_local: other threads of the process (the RCCL watchdog of a data-parallel job polls events) may keep
        # making HIP calls while this thread captures
        gc_was_on = gc.isenabled()
        gc.disable()      # (no collection, hence no finalizer of some unrelated object, between the capture's begin and end)
        from . import hip_adam
        self._pinned = []      # pinned table buffers the captured optimizer step copies from: they live as long as this graph
        hip_adam._capture_hosts = self._pinned
        from . import hip_batchnorm
        self._frozen_consts = []      # cached inference constants a frozen-weights graph reads (hip_batchnorm.frozen_capture)
        frozen = hip_batchnorm.frozen_capture(self._frozen_consts) if not self.weights_change else contextlib.nullcontext()
        try:
            with frozen, torch.cuda.graph(graph, capture_error_mode="thread_local"):
                out = self.fn()
        finally:
            hip_adam._capture_hosts = None
            if gc_was_on:
                gc.enable()
        torch.cuda.synchronize()
        # the rebuild of the weight packs captured at the head of the step covers EVERY registered layer of the process: keep what
        # it reads and writes alive as long as this graph may be replayed (hip_linear.registered_packs)
        self._pack_refs = hip_linear.registered_packs(torch.device("cuda", torch.cuda.current_device()))
        # A pack first created INSIDE the capture is cached as valid for the current epoch although its pack kernel was only
        # recorded, never executed: an eager forward between capture() and the first replay would read uninitialised memory.
        # Stale again after the capture, whatever the step did (a captured optimizer step bumps the epoch itself).
        if self.weights_change:
            hip_linear.invalidate_weight_packs()
        _bury()
        raw = graph.raw_cuda_graph()
        L = lib()
        handle = ctypes.c_void_p(0)
        check(L.nsdp_graph_exec_create(ctypes.c_void_p(int(raw)), ctypes.c_int(self.max_streams), ctypes.byref(handle)),
              "nsdp_graph_exec_create")
        vals = [ctypes.c_int(0) for _ in range(5)]
        check(L.nsdp_graph_exec_info(handle, *[ctypes.byref(v) for v in vals]), "nsdp_graph_exec_info")
        self.info = dict(zip(("nodes", "kernels", "streams", "cross_stream_edges", "own_graph_nodes"), (v.value for v in vals)))
        self._graph, self._handle, self._out = graph, handle, out
        return self

    def __call__(self):
        if self._graph is None:
            raise RuntimeError("capture() first")
        check(lib().nsdp_graph_exec_launch(self._handle, stream_ptr()), "nsdp_graph_exec_launch")
        # A replay runs no Python: the optimizer's post-step hook (which advances the weights epoch in the eager loop) does
        # not fire, while the packs -- rebuilt at the HEAD of the replayed step -- are one optimizer step older than the
        # parameters afterwards.  Declare them stale, so that the next EAGER forward (validate_on_batch, an odd-shape batch
        # of GraphedTrainOnBatch) repacks; the next replay repacks anyway.  One integer increment.
        if self.weights_change:
            from . import hip_batchnorm, hip_linear
            hip_linear.invalidate_weight_packs()
            hip_batchnorm.invalidate_inference_constants()      # (the replayed norms rewrote their running statistics)
        return self._out

    def timed_replay(self, name_substr: str):
        """One replay with HIP events around every kernel whose name contains ``name_substr``: (launches, total ms) as they
        run INSIDE the replay, beside the other streams' kernels.  Synchronises; a measurement, not for timed regions."""
        if self._graph is None:
            raise RuntimeError("capture() first")
        n, ms = ctypes.c_longlong(0), ctypes.c_double(0.0)
        check(lib().nsdp_graph_exec_launch_timed(self._handle, stream_ptr(), name_substr.encode(), ctypes.byref(n),
                                                 ctypes.byref(ms)), "nsdp_graph_exec_launch_timed")
        if self.weights_change:
            from . import hip_batchnorm, hip_linear
            hip_linear.invalidate_weight_packs()
            hip_batchnorm.invalidate_inference_constants()
        return int(n.value), float(ms.value)

    def close(self):
        # Destroying a graph (its executor's streams and events, the torch graph and its memory pool) is not something a
        # thread may do while it CAPTURES another one -- and Python's cycle collector runs `__del__` whenever it likes, in
        # the middle of a capture too (a GraphedTrainOnBatch and the lambda of its step form a cycle): the process aborted
        # there.  While a capture is in progress the remains are parked and freed at the next safe point.
        if self._handle or self._graph is not None:
            _graveyard.append((self._handle, self._graph, getattr(self, "_pinned", None)))
            self._handle = ctypes.c_void_p(0)
            self._graph = None
            self._pinned = None
            self._pack_refs = None
        _bury()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _geometry_tensors(g, out=None, seen=None):
    """The tensors of a model.geometry() bundle in a fixed order (each once; the query points -- an INPUT -- stay out)."""
    out = [] if out is None else out
    seen = set() if seen is None else seen
    if torch.is_tensor(g):
        if id(g) not in seen:
            seen.add(id(g))
            out.append(g)
    elif isinstance(g, dict):
        for k in sorted(g):
            if k != "query_points":
                _geometry_tensors(g[k], out, seen)
    elif isinstance(g, (list, tuple)):
        for v in g:
            _geometry_tensors(v, out, seen)
    return out


class PipelinedGeometry:
    """The NEXT batch's index sets computed beside the CURRENT step.

    Farthest-point sampling is a chain of ~600 dependent iterations and the neighbour searches wait for its centres: ~1 ms that
    no batch size shortens and that a step which searches for itself has at the head of its critical chain (with the chip mostly
    idle: one workgroup per shape).  None of it depends on a parameter -- it is a function of the batch.  A loop that knows its
    next batch (every loader does) therefore runs ``model.geometry(next batch)`` on a stream of its own under the current
    step's forward pass, and the next step starts with its index sets in place: the same searches, once per batch, the same
    results bit for bit; only WHEN they run changes.  Measured (profiles/r5_small_wgrad_and_sweeps.txt): the evaluation pass at
    B = 8 4.42 -> 4.24 ms, the train step 13.62 -> 13.37 ms at B = 8 and 39.12 -> 38.95 ms at B = 32 -- the searches still cost
    their chip time (a timing-only ablation that SKIPS them is worth 0.44 ms at B = 8), only the sampling chain's latency leaves
    the critical path.  Opt-in.

        pipe = PipelinedGeometry(model, inputs=lambda d: (d["space_samples_src"], d["surface_samples_inputs"]))
        pipe.prime(static_batch, training=True)                  # eager, once: the first batch's sets
        def fn():                                                # the function a GraphedStep captures
            pipe.prefetch(static_next_batch)                     #   next batch's sets, on the geometry stream
            loss = step(static_batch, geometry=pipe.current)     #   this batch's step, searching nothing
            pipe.rotate()                                        #   next -> current (device copies of the index tensors)
            return loss

    The caller keeps the invariant `pipe.current` == geometry of what the step reads: it writes batch i + 1 into the static
    "next" tensors before replay i and batch i + 1 into the static "current" ones before replay i + 1 (GraphedTrainOnBatch does
    this, and re-primes eagerly whenever a batch arrives that was not announced)."""

    def __init__(self, model, inputs):
        if not hasattr(model, "geometry"):
            raise TypeError("PipelinedGeometry: the model has no geometry() method")
        self.model, self.inputs = model, inputs
        self.current = self._next = None
        self.training = True
        self._stream = None

    def prime(self, batch, training=True):
        """Geometry of ``batch`` (the static tensors the step reads), eagerly, on the current stream.  The first call creates the
        `current` buffers; later calls refill them in place (a replay reads the same addresses)."""
        self.training = bool(training)
        g = self.model.geometry(*self.inputs(batch), training=self.training)
        if self.current is None:
            self.current = g
        else:
            cur, new = _geometry_tensors(self.current), _geometry_tensors(g)
            if len(cur) != len(new) or any(a.shape != b.shape or a.dtype != b.dtype for a, b in zip(cur, new)):
                raise RuntimeError("PipelinedGeometry: this batch's index sets do not have the shapes of the buffers in place")
            for a, b in zip(cur, new):
                a.copy_(b)
        return self.current

    def prefetch(self, next_batch):
        dev = _geometry_tensors(self.current)[0].device
        main = torch.cuda.current_stream(dev)
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=dev)
            self._tick = torch.zeros(1, dtype=torch.int32, device=dev)
        # one trivial launch on the step's own stream first, so that the search is a fork of the step and not a root of its own:
        # the graph executor then gives the step (the longer chain behind the fork) the main stream and the search the side
        # stream, which it shares with the weight gradients of the backward pass -- idle until then
        self._tick.add_(1)
        self._stream.wait_stream(main)
        with torch.cuda.stream(self._stream):
            self._next = self.model.geometry(*self.inputs(next_batch), training=self.training)

    def rotate(self):
        cur, nxt = _geometry_tensors(self.current), _geometry_tensors(self._next)
        if len(cur) != len(nxt) or any(a.shape != b.shape or a.dtype != b.dtype for a, b in zip(cur, nxt)):
            raise RuntimeError("PipelinedGeometry: the next batch's index sets do not have the current batch's shapes")
        main = torch.cuda.current_stream(cur[0].device)
        main.wait_stream(self._stream)
        for a, b in zip(cur, nxt):
            # a kernel node (a memcpy node is replayed by the executor as a graph of its own), on the BITS: an integer add of 0
            # -- a float add would turn -0.0 into +0.0 in the coordinate tensors
            if a.dtype.is_floating_point and a.element_size() == 4:
                torch.add(b.view(torch.int32), 0, out=a.view(torch.int32))
            else:
                torch.add(b, 0, out=a)
            b.record_stream(main)
        self._next = None


class GraphedTrainOnBatch:
    """Drop-in for the reference-shaped ``train_on_batch(model, optimizer, data_dict, config) -> float`` of
    nsdp_amd.model (reference model/deformation_networks.py:63-77, model/flow_arbitrary.py:30-48): the first call runs
    eagerly, the second one captures the step (its ``tensor_step`` form: the same statements without ``loss.item()``) over
    static copies of the batch, and from then on every call is a copy of the batch into the static tensors plus one replay
    -- the sequence of optimizer steps is exactly the eager loop's.  A batch of other shapes (the last, shorter one of an epoch) runs eagerly.  Returns the loss as a float,
    like the reference -- that read-back is the one host sync per step the reference has as well.

    ``reducer`` (nsdp_amd.parallel.GradAllReducer): this process is one rank of a data-parallel job.  A collective cannot be
    captured, so the step becomes one graph per side of each collective -- [zero the flat gradient, forward, loss, the
    decoder's backward] | all-reduce of bucket 0, asynchronous | [the encoder's backward] | all-reduce of bucket 1, wait for
    both | [optimizer.step] -- with the RCCL calls enqueued eagerly between the replays: the decoder's gradients travel under
    the encoder's backward pass.  Eager steps (the first one, odd shapes) run the same two-pass backward
    (GradAllReducer.backward).  The returned loss is this rank's."""

    def __init__(self, train_on_batch, max_streams: int | None = None, reducer=None, pipeline_geometry=None, overlap=None):
        if not hasattr(train_on_batch, "tensor_step"):
            raise TypeError("train_on_batch has no `tensor_step` form (the step without its loss.item())")
        if reducer is not None and not hasattr(train_on_batch, "loss_fn"):
            raise TypeError("train_on_batch has no `loss_fn` form (forward + loss), needed to split the step around the exchange")
        self.eager = train_on_batch
        self.max_streams = max_streams
        self.reducer = reducer
        self.exchanges_gradients = reducer is not None      # (nsdp_amd.train.fit: do not wrap me again)
        # ``overlap``: the decoder bucket's all-reduce under the encoder's backward pass (two autograd passes).  None = NSDP_DP_OVERLAP
        # or "auto": on for eager steps, OFF for the captured step -- each graph boundary joins the executor's streams, and the
        # head / tail boundary costs the overlap of the decoder's weight gradients with the encoder's backward chain (measured at
        # one rank, B = 32: 38.2-38.3 ms plain, 38.75-38.9 with two graphs, 39.5-39.7 with three; the exchange itself is 0.05-0.2 ms)
        mode = os.environ.get("NSDP_DP_OVERLAP", "auto") if overlap is None else ("on" if overlap else "off")
        self.overlap_eager, self.overlap_graph = mode != "off", mode == "on"
        # ``pipeline_geometry``: a function data_dict -> (points, surface inputs) -- the replayed step then takes its index sets
        # from a PipelinedGeometry and computes the NEXT batch's (``next_data_dict=`` of the call) beside itself.  A batch that
        # was not announced that way costs one eager geometry pass before its replay; results never depend on it.
        import inspect
        if pipeline_geometry is not None and "geometry" not in inspect.signature(train_on_batch.tensor_step).parameters:
            pipeline_geometry = None      # (a step function that does not take its index sets from outside: FlowArbitrary)
        self.pipeline_inputs = pipeline_geometry
        self.accepts_next_batch = pipeline_geometry is not None      # (nsdp_amd.train.fit: look one batch ahead)
        self._pipe = self._static_next = self._announced = None
        self._shapes = None
        self._static = None
        self._step = None
        self._tail = None
        self._update = None
        self.replays = self.eager_calls = 0

    @staticmethod
    def _sig(data_dict):
        return tuple(sorted((k, tuple(v.shape), v.dtype) for k, v in data_dict.items() if torch.is_tensor(v)))

    def _eager_step(self, model, optimizer, data_dict, config):
        self.eager_calls += 1
        if self.reducer is None:
            return float(self.eager.tensor_step(model, optimizer, data_dict, config))
        red = self.reducer
        red.zero_grad(two_pass=self.overlap_eager)
        loss = self.eager.loss_fn(model, data_dict, config)
        red.backward(loss)
        red.finish()
        optimizer.step()
        return float(loss)

    @staticmethod
    def _same_batch(announced, data_dict):
        return (announced is not None and announced.keys() == {k for k, v in data_dict.items() if torch.is_tensor(v)}
                and all(data_dict[k] is t and t._version == ver for k, (t, ver) in announced.items()))

    def __call__(self, model, optimizer, data_dict, config, next_data_dict=None):
        sig = self._sig(data_dict)
        if self._shapes is None:
            # The very first step runs eagerly (with the optimizer already in its capturable form): it creates the optimizer
            # state.  State created INSIDE a capture would be re-initialised by every replay (Adam's moments zeroed each
            # step) -- and an extra warm-up step would change what the loop computes.
            capturable_adam(optimizer)
            self._shapes = sig
            return self._eager_step(model, optimizer, data_dict, config)
        if sig != self._shapes:
            return self._eager_step(model, optimizer, data_dict, config)
        if self._step is None:
            self._static = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in data_dict.items()}
            piped = self.pipeline_inputs is not None and hasattr(model, "geometry")
            if piped:
                self._static_next = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in data_dict.items()}
                self._pipe = PipelinedGeometry(model, self.pipeline_inputs)
                self._pipe.prime(self._static, training=True)
                self._announced = {k: (v, v._version) for k, v in data_dict.items() if torch.is_tensor(v)}

            streams = self.max_streams

            def piped_fn(step):      # next batch's index sets beside the step, then next -> current
                if not piped:
                    return step(None)
                self._pipe.prefetch(self._static_next)
                out = step(self._pipe.current)
                self._pipe.rotate()
                return out
            if self.reducer is None:
                def fn():
                    return piped_fn(lambda g: self.eager.tensor_step(model, optimizer, self._static, config, **({"geometry": g} if g is not None else {})))
                self._step = GraphedStep(fn, streams).capture(warmup=0)      # (a capture executes nothing)
            else:
                red = self.reducer

                def head():
                    red.zero_grad(two_pass=self.overlap_graph)
                    # (the search of the next batch's index sets forks a stream that its hand-over joins: both in ONE graph --
                    # beside the forward pass in the two-graph form, beside the encoder's backward in the tail of the three-graph one)
                    if piped and not (self.overlap_graph and red._can_cut):
                        self._pipe.prefetch(self._static_next)
                    g = self._pipe.current if piped else None
                    loss = self.eager.loss_fn(model, self._static, config, **({"geometry": g} if g is not None else {}))
                    if not red.backward_head(loss):      # (no cut: the whole backward here, no tail graph)
                        loss.backward()
                        if piped:
                            self._pipe.rotate()
                    return loss

                def tail():
                    if piped:
                        self._pipe.prefetch(self._static_next)
                    red.backward_tail()
                    if piped:      # (the hand-over overwrites the index sets the backward pass reads: backward first)
                        self._pipe.rotate()
                    return red.flat
                self._step = GraphedStep(head, streams).capture(warmup=0)
                if red._root_grads is not None:      # the forward was cut: the encoder's backward is a graph of its own
                    red.start(0)
                    # (the tail rebuilds no weight pack -- it runs on what the head's forward saved: a frozen-weights capture)
                    self._tail = GraphedStep(tail, streams, weights_change=False).capture(warmup=0)
                red.finish()
                self._update = GraphedStep(lambda: optimizer.step(), self.max_streams).capture(warmup=0)
        announced = self._pipe is not None and self._same_batch(self._announced, data_dict)
        for k, v in data_dict.items():
            if torch.is_tensor(v):
                # (an announced batch was copied into the static "next" tensors when it was announced, and its index sets were
                # computed from THAT copy: the step reads the same bytes, whatever happened to the caller's tensors since)
                self._static[k].copy_(self._static_next[k] if announced else v, non_blocking=True)
        if self._pipe is not None:
            if not announced:
                self._pipe.prime(self._static, training=True)      # not the batch the last replay prepared: search now, eagerly
                self.unannounced = getattr(self, "unannounced", 0) + 1
            self._announced = None
            if next_data_dict is not None and self._sig(next_data_dict) == self._shapes:
                for k, v in next_data_dict.items():
                    if torch.is_tensor(v):
                        self._static_next[k].copy_(v, non_blocking=True)
                self._announced = {k: (v, v._version) for k, v in next_data_dict.items() if torch.is_tensor(v)}
        if self._pipe is not None and os.environ.get("NSDP_PIPE_VERIFY") == "1":      # (debug: is `current` this batch's geometry?)
            torch.cuda.synchronize()
            fresh = _geometry_tensors(model.geometry(*self.pipeline_inputs(self._static), training=True))
            for i, (a, b) in enumerate(zip(_geometry_tensors(self._pipe.current), fresh)):
                if not torch.equal(a, b):
                    same_in = all(torch.equal(self._static[k], self._static_next[k]) for k in self._static if torch.is_tensor(self._static[k]))
                    again = _geometry_tensors(model.geometry(*self.pipeline_inputs(self._static_next), training=True))[i]
                    raise AssertionError(f"pipelined geometry: tensor {i} {tuple(a.shape)} {a.dtype} differs before replay {self.replays}: "
                                         f"{int((a != b).sum())} of {a.numel()} elements; static == static_next: {same_in}; "
                                         f"geometry(static_next) == current: {torch.equal(again, a)}; == fresh: {torch.equal(again, b)}; "
                                         f"current[:8] {a.flatten()[:8].tolist()} fresh[:8] {b.flatten()[:8].tolist()}")
        self.replays += 1
        loss = self._step()
        if self.reducer is not None:
            if self._tail is not None:
                self.reducer.start(0)      # the decoder's gradients: under the encoder's backward
                self._tail()
            self.reducer.finish()          # bucket 1 (or both), then the wait
            self._update()
        return float(loss)
