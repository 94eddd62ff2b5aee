"""The multi-stream graph executor (csrc/graph_exec.hip, nsdp_amd/graph_step.py): a captured TDNet train step replayed
from C must train exactly like the eager step -- same losses step for step, same weights afterwards."""
import pytest
import torch

from helpers import build_product, model_cfg, restore_model, snapshot_model, to_dev
from nsdp_amd import synth

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _make(cfg, seed, data):
    from nsdp_amd.model import optimizer_factory
    from nsdp_amd.model.utils import compute_l2_error
    model, _, _ = build_product(cfg, seed, DEV)
    model.train()
    _, opt = optimizer_factory({"optimizer": "Adam", "lr": 5e-5}, model.parameters())

    def step():
        opt.zero_grad(set_to_none=True)
        pred = model(data["space_samples_src"], data["surface_samples_inputs"])
        loss = compute_l2_error(pred, data["space_samples_tgt"])
        loss.backward()
        opt.step()
        return loss
    return model, opt, step


@pytest.mark.parametrize("B,npl,ns,nq,streams", [(2, [256, 64, 16], 256, 128, 4), (16, [2048, 500, 100], 2048, 8192, 4),
                                                  (2, [256, 64, 16], 256, 128, 1)])
def test_replayed_step_trains_like_the_eager_step(B, npl, ns, nq, streams):
    """B = 16 at full point counts takes the weight-gradient side stream (131 072 output rows): the captured graph has
    cross-stream edges there, which the executor turns into events between its own streams."""
    from nsdp_amd.graph_step import GraphedStep, capturable_adam
    cfg = model_cfg("forward", npl)
    data = to_dev(synth.make_batch(91, B, ns, nq), DEV)
    model_e, opt_e, step_e = _make(cfg, 91, data)
    eager = [float(step_e()) for _ in range(7)]
    model_g, opt_g, step_g = _make(cfg, 91, data)
    capturable_adam(opt_g)
    gs = GraphedStep(step_g, max_streams=streams).capture(warmup=3)
    assert gs.info["kernels"] > 300 and gs.info["streams"] <= streams
    from nsdp_amd import hip_linear
    side = hip_linear._PARAM_GRADS_DIRECT and hip_linear._OVERLAP_WGRAD is not False      # (knobs that keep one stream)
    if streams > 1 and side:      # (a captured step takes the weight-gradient side stream at every batch size)
        assert gs.info["streams"] >= 2 and gs.info["cross_stream_edges"] >= 50, gs.info
    got = [float(gs()) for _ in range(4)]
    torch.cuda.synchronize()
    for a, b in zip(eager[3:], got):
        assert abs(a - b) <= 2e-3 * abs(a) + 1e-7, (eager, got)
    # (Adam moves a weight by ~lr per step whatever the gradient's size: a gradient that is analytically zero -- the bias
    # of a conv in front of a train-mode BatchNorm -- is rounding noise whose sign differs between two runs, so such an
    # entry may sit anywhere within +- steps x lr; the losses above are the sharp check, this one catches a lost update)
    for (k, p), (_, q) in zip(model_e.named_parameters(), model_g.named_parameters()):
        assert float((p - q).abs().max()) <= 2 * 7 * 5e-5, k
    # new inputs go in through the static tensors
    data["space_samples_tgt"].add_(0.25)
    moved = float(gs())
    torch.cuda.synchronize()
    assert moved > 1.5 * got[-1]
    gs.close()


@pytest.mark.parametrize("B,npl,ns,nq", [(2, [256, 64, 16], 256, 128), (16, [2048, 500, 100], 2048, 8192)])
def test_replayed_step_is_bit_equal_to_the_eager_step(B, npl, ns, nq):
    """The train step has no floating-point atomics left (decoder anchor tables: scatter as a GEMM; global token: ordered
    partial sums; encoder: inverse neighbour lists; weight gradients: ordered partial sums), so its result does not depend
    on WHEN a kernel runs -- only on whether it ran after its inputs were complete.  That makes bit-equality the race
    detector of the replay: the captured step replayed on 1, 2 and 4 streams (csrc/graph_exec.hip: own topological order,
    chain decomposition, events on the cross-stream edges) must reproduce the eager step's loss, EVERY gradient, every
    updated weight and every BatchNorm buffer bit for bit, for two consecutive steps.  A dropped cross-stream wait shows up
    as a gradient computed from a half-written tensor.  B = 16 at full point counts is the shape class bench.py times
    (weight gradients on the side stream, 8-wave bf16x3 GEMMs, one-hot scatters)."""
    from helpers import nondeterministic_knobs
    if nondeterministic_knobs():
        pytest.skip("the step is not bit-reproducible under " + ", ".join(nondeterministic_knobs()))
    from nsdp_amd.graph_step import GraphedStep, capturable_adam
    cfg = model_cfg("forward", npl)
    data = to_dev(synth.make_batch(93, B, ns, nq), DEV)
    model, opt, step = _make(cfg, 93, data)
    capturable_adam(opt)
    snap = snapshot_model(model)
    step()                                     # creates the optimizer state (a capture must find it in place)
    torch.cuda.synchronize()

    def two_steps(run):
        restore_model(model, snap, opt)
        out = []
        for _ in range(2):
            loss = run()
            torch.cuda.synchronize()
            out.append({"loss": loss.detach().clone(),
                        "grads": {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None},
                        "state": snapshot_model(model)})
        return out

    def same(a, b, what):
        bad = []
        for i, (x, y) in enumerate(zip(a, b)):
            if not torch.equal(x["loss"], y["loss"]):
                bad.append(f"step {i} loss {float(x['loss'])!r} vs {float(y['loss'])!r}")
            assert x["grads"].keys() == y["grads"].keys()
            bad += [f"step {i} grad {k}" for k in x["grads"] if not torch.equal(x["grads"][k], y["grads"][k])]
            bad += [f"step {i} state {k}" for k in x["state"] if not torch.equal(x["state"][k], y["state"][k])]
        assert not bad, f"{what}: {len(bad)} tensors differ, first: {bad[:8]}"

    eager = two_steps(step)
    same(eager, two_steps(step), "eager step run twice (is the step deterministic at all?)")
    for streams in (1, 2, 4):
        restore_model(model, snap, opt)
        gs = GraphedStep(step, max_streams=streams).capture(warmup=0)      # (a capture executes nothing)
        if streams > 1:
            assert gs.info["streams"] >= 2 and gs.info["cross_stream_edges"] >= 2, gs.info
        same(eager, two_steps(gs), f"replay on {streams} stream(s) vs eager")
        gs.close()


def _make_any(cfg, seed, data, lr=5e-5):
    """(model, optimizer, step) for either model type: the step is the reference's train_on_batch without the host read-back
    (train_fn.tensor_step: model/deformation_networks.py:63-77, model/flow_arbitrary.py:30-48)."""
    from nsdp_amd.model import optimizer_factory
    model, train_fn, _ = build_product(cfg, seed, DEV)
    model.train()
    _, opt = optimizer_factory({"optimizer": "Adam", "lr": lr}, model.parameters())
    return model, opt, (lambda: train_fn.tensor_step(model, opt, data, cfg))


@pytest.mark.parametrize("mtype,dtype,B,npl,ns,nq", [
    ("arbitrary", "f32", 2, [256, 64, 16], 256, 128),            # FlowArbitrary, tiny
    ("arbitrary", "f32", 8, [2048, 500, 100], 2048, 8192),       # FlowArbitrary at full point counts (config 3's function)
    ("forward", "bf16", 2, [256, 64, 16], 256, 128),             # bf16 storage, tiny
    ("forward", "bf16", 16, [2048, 500, 100], 2048, 8192),       # bf16 storage at the shape class bench.py --dtype bf16 times
    ("arbitrary", "bf16", 2, [256, 64, 16], 256, 128),
    ("arbitrary", "bf16", 8, [2048, 500, 100], 2048, 8192),      # what bench.py --workload arbitrary_train --dtype bf16 replays
])
def test_replayed_arbitrary_and_bf16_steps_are_bit_equal_to_the_eager_step(mtype, dtype, B, npl, ns, nq, monkeypatch):
    """The launcher of BASELINE config 3 under the same race detector as the fp32 forward model above: a FlowArbitrary step
    (reference model/flow_arbitrary.py:15-48: two networks, ONE encoder pass per cloud here, BatchNorm buffers updated twice
    per step -- `num_batches_tracked` advances by 2 --, a coordinate-gradient path through the second network's geometry) and
    the bf16-storage step, captured and replayed on 1 / 2 / 4 streams, must reproduce the eager step's loss, EVERY gradient,
    weight and BatchNorm buffer bit for bit over two steps.
    The eager step takes the weight-gradient side stream here as a captured step always does (NSDP_WGRAD_STREAM=1): below
    131 072 output rows the eager launcher would keep the weight gradients on the main stream, where their partial sums are
    split over all 256 compute units instead of 256 - SIDE_RESERVE_CUS -- another, equally valid rounding (measured at B = 8:
    188 of 521 gradients differ, by <= 2e-9 absolute; no race: with the same stream decision the two are equal)."""
    from helpers import nondeterministic_knobs
    if nondeterministic_knobs():
        pytest.skip("the step is not bit-reproducible under " + ", ".join(nondeterministic_knobs()))
    from nsdp_amd import hip_linear, precision
    if hip_linear._OVERLAP_WGRAD == "auto":      # (NSDP_WGRAD_STREAM=0 / 1: eager and captured steps decide alike anyway)
        monkeypatch.setattr(hip_linear, "_OVERLAP_WGRAD", True)
    from nsdp_amd.graph_step import GraphedStep, capturable_adam
    cfg = model_cfg(mtype, npl)
    data = to_dev(synth.make_batch(193, B, ns, nq), DEV)
    with precision.storage(dtype):
        model, opt, step = _make_any(cfg, 193, data)
        capturable_adam(opt)
        snap = snapshot_model(model)
        step()                                     # creates the optimizer state (a capture must find it in place)
        torch.cuda.synchronize()

        def two_steps(run):
            restore_model(model, snap, opt)
            out = []
            for _ in range(2):
                loss = run()
                torch.cuda.synchronize()
                out.append({"loss": loss.detach().clone(),
                            "grads": {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None},
                            "state": snapshot_model(model)})
            return out

        def diff(a, b):
            bad = []
            for i, (x, y) in enumerate(zip(a, b)):
                if not torch.equal(x["loss"], y["loss"]):
                    bad.append(f"step {i} loss {float(x['loss'])!r} vs {float(y['loss'])!r}")
                assert x["grads"].keys() == y["grads"].keys()
                bad += [f"step {i} grad {k}" for k in x["grads"] if not torch.equal(x["grads"][k], y["grads"][k])]
                bad += [f"step {i} state {k}" for k in x["state"] if not torch.equal(x["state"][k], y["state"][k])]
            return bad

        eager = two_steps(step)
        if mtype == "arbitrary":      # every BatchNorm of a network that runs twice per step counts two batches per step
            nbt = [int(v) - int(snap[k]) for k, v in eager[0]["state"].items() if k.endswith("num_batches_tracked")]
            assert nbt and set(nbt) <= {1, 2} and 2 in nbt, sorted(set(nbt))
        again = diff(eager, two_steps(step))
        assert not again, f"the eager {mtype} / {dtype} step is not deterministic: {len(again)} tensors differ, first {again[:6]}"
        for streams in (1, 2, 4):
            restore_model(model, snap, opt)
            gs = GraphedStep(step, max_streams=streams).capture(warmup=0)      # (a capture executes nothing)
            if streams > 1:
                assert gs.info["streams"] >= 2 and gs.info["cross_stream_edges"] >= 2, gs.info
            bad = diff(eager, two_steps(gs))
            assert not bad, f"replay on {streams} stream(s) vs eager: {len(bad)} tensors differ, first: {bad[:8]}"
            gs.close()


def test_an_eval_step_captured_with_changing_weights_follows_the_training_in_between():
    """GraphedStep(weights_change=True) over an EVAL forward is replay-safe while training continues: weight packs are rebuilt
    at the head of every replay, and the BatchNorm inference constant 1 / sqrt(running_var + eps) must be a NODE of the graph,
    not the tensor an earlier eager evaluation cached on the module (that tensor would be frozen into every replay next to
    the live running_mean, and freed under the graph by the next eager evaluation).  Sequence: eager eval (fills the caches),
    capture, train steps, replay == eager eval at the new weights and statistics, bit for bit; again after more training."""
    from nsdp_amd.graph_step import GraphedStep
    from nsdp_amd.model import optimizer_factory
    cfg = model_cfg("forward", [256, 64, 16])
    data = to_dev(synth.make_batch(94, 2, 256, 128), DEV)
    model, train_fn, _ = build_product(cfg, 94, DEV)
    _, opt = optimizer_factory({"optimizer": "Adam", "lr": 5e-4}, model.parameters())

    def evaluate():
        with torch.no_grad():
            return model(data["space_samples_src"], data["surface_samples_inputs"])
    model.eval()
    first = evaluate().clone()
    gs = GraphedStep(evaluate, weights_change=True).capture(warmup=1)
    for rounds in range(2):
        model.train()
        for _ in range(3):
            train_fn(model, opt, data, cfg)
        model.eval()
        got = gs().clone()
        torch.cuda.synchronize()
        want = evaluate()
        torch.cuda.synchronize()
        assert bool(torch.isfinite(want).all()) and bool(torch.isfinite(got).all()), (bool(torch.isfinite(want).all()), bool(torch.isfinite(got).all()))
        assert not torch.equal(want, first)                     # (training moved the weights and the running statistics)
        assert torch.equal(got, want), float((got - want).abs().max())
        evaluate()                                              # (an eager evaluation replaces the modules' cached constants)
    gs.close()


def test_set_lr_changes_the_update_of_a_replayed_step():
    from nsdp_amd.graph_step import GraphedStep, capturable_adam, set_lr
    cfg = model_cfg("forward", [256, 64, 16])
    data = to_dev(synth.make_batch(92, 2, 256, 128), DEV)
    model, opt, step = _make(cfg, 92, data)
    capturable_adam(opt)
    gs = GraphedStep(step).capture(warmup=1)
    w = next(model.decoder.parameters())
    w0 = w.detach().clone()
    gs()
    d1 = float((w.detach() - w0).abs().max())
    set_lr(opt, 5e-3)                       # 100x
    w1 = w.detach().clone()
    gs()
    d2 = float((w.detach() - w1).abs().max())
    torch.cuda.synchronize()
    assert d2 > 20 * d1 > 0, (d1, d2)


def test_optimizer_state_created_inside_a_capture_is_reset_by_every_replay():
    """Why capture() warms up / GraphedTrainOnBatch runs its first step eagerly: documented behaviour of stream capture."""
    from nsdp_amd.graph_step import GraphedStep, capturable_adam
    p = torch.nn.Parameter(torch.ones(1024, device=DEV))
    opt = torch.optim.Adam([p], lr=1e-2)
    capturable_adam(opt)

    def step():
        opt.zero_grad(set_to_none=True)
        (p * p).sum().backward()
        opt.step()
        return p.detach().sum()
    gs = GraphedStep(step).capture(warmup=0)
    for _ in range(3):
        gs()
    torch.cuda.synchronize()
    assert float(opt.state[p]["step"]) == 1.0          # re-initialised and incremented once per replay, never beyond 1
    gs.close()


def test_replays_interleaved_with_validation_and_odd_shape_batches_equal_the_eager_loop():
    """GraphedTrainOnBatch in the shape of a real epoch: replays, a validation pass in between (an EAGER forward right after
    a replay: the weight packs a replay leaves behind are one optimizer step old and must not be taken for current), a batch
    of another shape (eager fallback: its gradient must be taken at the current weights), replays again.  The step is
    deterministic, so the whole sequence -- train losses, validation losses, final weights -- must equal the plain eager
    loop's bit for bit."""
    from helpers import nondeterministic_knobs
    if nondeterministic_knobs():
        pytest.skip("the step is not bit-reproducible under " + ", ".join(nondeterministic_knobs()))
    from nsdp_amd.graph_step import GraphedTrainOnBatch, capturable_adam
    from nsdp_amd.model import optimizer_factory
    from nsdp_amd.model.deformation_networks import validate_on_batch_with_cano as val_fn
    cfg = model_cfg("forward", [256, 64, 16])
    main = to_dev(synth.make_batch(95, 2, 256, 128), DEV)
    other = to_dev(synth.make_batch(96, 2, 256, 128), DEV)
    odd = to_dev(synth.make_batch(97, 1, 256, 128), DEV)
    script = [("train", main), ("train", other), ("train", main), ("val", other), ("train", odd), ("train", other),
              ("val", main), ("train", main), ("val", odd)]

    def run(graphed):
        model, train_fn, _ = build_product(cfg, 95, DEV)
        _, opt = optimizer_factory({"optimizer": "Adam", "lr": 5e-5}, model.parameters())
        capturable_adam(opt)
        fn = GraphedTrainOnBatch(train_fn) if graphed else train_fn
        out = []
        for what, batch in script:
            if what == "train":
                model.train()
                out.append(fn(model, opt, batch, cfg))
            else:
                model.eval()
                out.append(val_fn(model, batch, cfg))
        torch.cuda.synchronize()
        if graphed:
            assert fn.replays == 4 and fn.eager_calls == 2, (fn.replays, fn.eager_calls)
        return out, {k: v.detach().clone() for k, v in model.state_dict().items()}

    e_out, e_state = run(False)
    g_out, g_state = run(True)
    assert e_out == g_out, (e_out, g_out)
    bad = [k for k in e_state if not torch.equal(e_state[k], g_state[k])]
    assert not bad, bad[:8]


def test_a_step_closed_while_another_one_is_captured_is_freed_afterwards():
    """Python's cycle collector may finalize a forgotten GraphedStep in the middle of another capture (it did: the process
    aborted inside hipGraphDestroy).  A close() during a capture parks the graph; it is destroyed after the capture."""
    from nsdp_amd import graph_step
    from nsdp_amd.graph_step import GraphedStep
    x = torch.ones(1024, device=DEV)
    y = torch.zeros(1024, device=DEV)
    old = GraphedStep(lambda: y.add_(x), weights_change=False).capture(warmup=0)
    old()

    def fn():
        y.mul_(2.0)
        old.close()                 # (what a finalizer run by the collector would do)
        assert graph_step._graveyard, "a close() during a capture must be deferred"
        return y
    new = GraphedStep(fn, weights_change=False).capture(warmup=0)
    assert not graph_step._graveyard
    new()
    torch.cuda.synchronize()
    assert float(y[0]) == 2.0
    new.close()


def test_timed_replay_measures_a_kernel_class_and_leaves_the_step_intact():
    """GraphedStep.timed_replay (bench.py's roofline.frac_replayed): a replay with event pairs around the kernels of one name --
    it counts the launches of that class, returns a positive duration, and is a replay like any other: the sequence
    replay, timed replay, replay equals three eager steps bit for bit."""
    from helpers import nondeterministic_knobs
    from nsdp_amd.graph_step import GraphedStep, capturable_adam
    cfg = model_cfg("forward", [256, 64, 16])
    data = to_dev(synth.make_batch(93, 2, 256, 128), DEV)
    model_e, opt_e, step_e = _make(cfg, 93, data)
    capturable_adam(opt_e)
    eager = [float(step_e()) for _ in range(4)]
    model_g, opt_g, step_g = _make(cfg, 93, data)
    capturable_adam(opt_g)
    first = float(step_g())                      # (creates the optimizer state eagerly, as GraphedTrainOnBatch does)
    gs = GraphedStep(step_g).capture(warmup=0)
    got = [first, float(gs())]
    n, ms = gs.timed_replay("linear_nt_kernel")
    got.append(float(gs._out))
    got.append(float(gs()))
    torch.cuda.synchronize()
    assert n > 20 and ms > 0.0, (n, ms)
    assert gs.timed_replay("no_such_kernel_name")[0] == 0
    if not nondeterministic_knobs():
        assert got == eager, (got, eager)
    gs.close()


def test_replays_leave_the_memory_of_a_model_that_died_after_the_capture_alone():
    """The rebuild of the weight packs at the head of a captured step is ONE launch over every registered layer of the process --
    another live model's too.  When that model dies after the capture, the replays must not go on packing its (freed) weights
    into its (freed) pack buffers: whoever owns that memory by then -- here canary tensors; in the pipelined harness it was the
    NEXT batch, resident while the step ran -- would be overwritten.  The graph keeps what it touches alive
    (hip_linear.registered_packs)."""
    import gc
    from nsdp_amd.graph_step import GraphedStep, capturable_adam
    cfg = model_cfg("forward", [256, 64, 16])
    data = to_dev(synth.make_batch(5, 2, 256, 128), DEV)
    other, _, other_step = _make(cfg, 3, data)
    other_step()                                   # registers the other model's packs
    model, opt, step = _make(cfg, 4, data)
    capturable_adam(opt)
    step()
    g = GraphedStep(step).capture(warmup=1)
    g()
    torch.cuda.synchronize()
    del other, other_step, _
    gc.collect()
    torch.cuda.synchronize()
    # an eager pass in between (a validation batch, an odd-shape batch): its pack rebuild drops the dead model's entries from the
    # registry -- from here on only the graph's own references keep those buffers
    with torch.no_grad():
        model(data["space_samples_src"], data["surface_samples_inputs"])
    torch.cuda.synchronize()
    # grab whatever the dead model released: many tensors of the sizes its weights and packs had
    canaries = [torch.full((n,), 12345.0, device=DEV) for n in (256, 1024, 4096, 16384, 65536, 262144) for _ in range(40)]
    for _ in range(3):
        g()
    torch.cuda.synchronize()
    bad = [i for i, c in enumerate(canaries) if not bool((c == 12345.0).all())]
    assert not bad, f"{len(bad)} of {len(canaries)} canary tensors were written by the replays"
    g.close()
