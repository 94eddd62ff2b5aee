"""GPU parity of the TDNet hot path against the golden fixtures (imported reference) and the oracle."""
import numpy as np
import pytest
import torch

from helpers import (batchnorm_three_launch, build_product, fixture_setup, l2_err, model_cfg, nondeterministic_knobs, restore_model,
                     run_forward, sample_flat, snapshot_model, to_dev)
from nsdp_amd import synth
from oracle import tdnet_ref

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
TOL_L2 = 1e-4  # north-star bar: <= 1e-4 L2 vs the reference CPU path, fp32


def _hooks(model, tape):
    from nsdp_amd.model.decoder.blocks import CrossTransformerBlock
    from nsdp_amd.model.encoder.blocks import ElementwiseMLP, TransformerBlock, TransformerSetAbstraction
    hs = []
    for name, mod in model.named_modules():
        if isinstance(mod, (TransformerBlock, ElementwiseMLP, CrossTransformerBlock)):
            hs.append(mod.register_forward_hook(lambda m, i, o, name=name: tape.__setitem__(name + ".out", o.detach())))
        elif isinstance(mod, TransformerSetAbstraction):
            hs.append(mod.register_forward_hook(lambda m, i, o, name=name: tape.__setitem__(name + ".out", o[1].detach())))
    return hs


@pytest.mark.parametrize("mtype", ["forward", "backward", "arbitrary"])
def test_eval_forward_matches_golden(mtype):
    fx, cfg, seed, data = fixture_setup("tiny_" + mtype, mtype)
    model, _, _ = build_product(cfg, seed, DEV)
    model.eval()
    from nsdp_amd import hip_decoder
    tape = {}
    hs = _hooks(model, tape)
    with torch.no_grad():
        out_fused = run_forward(model, cfg, to_dev(data, DEV))       # decoder = nsdp_decoder_fused_fwd
        hip_decoder.ENABLED = False                                    # layer-by-layer decoder: exposes the taps
        try:
            out = run_forward(model, cfg, to_dev(data, DEV))
        finally:
            hip_decoder.ENABLED = True
    for h in hs:
        h.remove()
    assert l2_err(out.cpu().numpy(), fx["eval_out"]) <= TOL_L2
    assert l2_err(out_fused.cpu().numpy(), fx["eval_out"]) <= TOL_L2
    checked = 0
    for key, ref in fx.items():
        if key.startswith("eval_tap/"):
            mine = tape[key[len("eval_tap/"):]]
            if mtype == "arbitrary" and key.startswith("eval_tap/model_canonicalize.decoder") and mine.shape[1] != int(fx["meta_ns"]):
                # the reference calls network 1 twice (space queries, then the surface samples: its tap holds the second call);
                # this library decodes [space queries ; surface samples] in one pass against one encoding
                assert mine.shape[1] == int(fx["meta_nq"]) + int(fx["meta_ns"])
                mine = mine[:, int(fx["meta_nq"]):].contiguous()
            mine = mine.cpu().numpy() if mine.numel() == ref.size and mine.dim() > 1 else sample_flat(mine, 64)
            np.testing.assert_allclose(mine.reshape(ref.shape), ref, rtol=0, atol=2e-4, err_msg=key)
            checked += 1
    assert checked >= 15


def test_full_shape_forward_matches_golden():
    """BASELINE configs[0]/[1] geometry: 2048 surface + 8192 query points, forward.yaml architecture."""
    fx, cfg, seed, data = fixture_setup("full_forward", "forward")
    model, _, _ = build_product(cfg, seed, DEV)
    model.eval()
    with torch.no_grad():
        out = run_forward(model, cfg, to_dev(data, DEV))
    # torch.argsort (the reference's kNN) is not stable: when the k-th and (k+1)-th anchor distances of a
    # query are bit-equal the reference's neighbour SET is arbitrary; such queries
    # (1 of 8192 here) are excluded -- everything else must meet the bar.
    from oracle import pointnet2_ref
    xyz0 = np.ascontiguousarray(data["surface_samples_inputs"][:, :, :3])
    xyz1 = np.take_along_axis(xyz0, fx["geo/fps1"][..., None].astype(np.int64), 1)
    xyz2 = np.take_along_axis(xyz1, fx["geo/fps2"][..., None].astype(np.int64), 1)
    _, d2 = pointnet2_ref.knn(data["space_samples_src"], xyz2, 8, return_dist=True)
    keep = d2[0, :, 6] != d2[0, :, 7]
    assert (~keep).sum() <= 2
    err = l2_err(out.cpu().numpy()[:, keep], fx["eval_out"][:, keep])
    assert err <= TOL_L2, err


def _variant_trace():
    """Context manager around nsdp_trace_*: the set of kernel template instances launched inside."""
    import contextlib
    import ctypes
    from nsdp_amd import _lib

    @contextlib.contextmanager
    def cm():
        L = _lib.lib()
        L.nsdp_trace_enable(1)
        names = set()
        try:
            yield names
        finally:
            L.nsdp_trace_enable(0)
            n = L.nsdp_trace_read(None, 0)
            buf = ctypes.create_string_buffer(n)
            L.nsdp_trace_read(buf, n)
            names.update(x for x in buf.value.decode().split("\n") if x)
    return cm()


def _check_train_step(fx, model, train_fn, cfg, data, chaotic_prefix=None, replay=False, loss_rtol=2e-5, chaotic_norm_rtol=5e-2,
                      chaotic_bn_rtol=2e-2):
    """One optimizer step of the product against what the imported reference produced for the same seeded inputs:
    loss, every gradient (norm + 16 samples), the None-gradient set, BatchNorm running statistics, Adam deltas.
    ``replay``: the checked step is a REPLAY of the captured step through the multi-stream graph executor
    (nsdp_amd.graph_step, what bench.py times) -- one eager step creates the optimizer state, weights / buffers / optimizer
    state are put back in place, the step is captured (nothing executes) and replayed once.
    ``chaotic_prefix``: parameters UPSTREAM of a discontinuous index selection (FlowArbitrary's first network: its output
    points are what the second network samples and groups) -- one neighbour that flips on a 1e-7 difference changes their
    gradient by percents while the loss moves by 1e-7; their norms are held to 5 % and their samples / Adam deltas are not
    compared entry by entry."""
    from nsdp_amd.model import optimizer_factory
    model.train()
    _, opt = optimizer_factory({"optimizer": "Adam", "lr": 5e-4, "lr_step": 200, "lr_decay": 0.1,
                                "weight_decay": 0.0}, model.parameters())
    before = {k: p.detach().clone() for k, p in model.named_parameters()}
    if replay:
        from helpers import restore_model, snapshot_model
        from nsdp_amd.graph_step import GraphedStep, capturable_adam
        capturable_adam(opt)
        dd = to_dev(data, DEV)
        snap = snapshot_model(model)
        train_fn.tensor_step(model, opt, dd, cfg)
        torch.cuda.synchronize()
        restore_model(model, snap, opt)
        gs = GraphedStep(lambda: train_fn.tensor_step(model, opt, dd, cfg)).capture(warmup=0)
        assert gs.info["kernels"] > 300, gs.info
        loss = float(gs())
        torch.cuda.synchronize()
    else:
        loss = train_fn(model, opt, to_dev(data, DEV), cfg)
    assert abs(loss - float(fx["train_loss"])) <= loss_rtol * max(1.0, abs(loss)), (loss, float(fx["train_loss"]))
    worst_norm = 0.0
    none = sorted(k for k, p in model.named_parameters() if p.grad is None)
    assert none == sorted(str(s) for s in fx["none_grads"])
    for k, p in model.named_parameters():
        if p.grad is None:
            continue
        gn = float(fx["grad_norm/" + k])
        mine = float(p.grad.double().norm())
        if chaotic_prefix and k.startswith(chaotic_prefix):
            worst_norm = max(worst_norm, (abs(mine - gn) - 2e-5) / max(gn, 1e-30))
            assert abs(mine - gn) <= chaotic_norm_rtol * gn + 2e-5, (k, mine, gn)
            continue
        # absolute floor: some gradients (e.g. fc_delta.2.bias in front of a train-mode BatchNorm) are
        # analytically ~0 and consist of fp32 cancellation noise whose value depends on summation order
        assert abs(mine - gn) <= 3e-3 * gn + 2e-5, (k, mine, gn)
        np.testing.assert_allclose(sample_flat(p.grad, 16), fx["grad_sample/" + k], rtol=5e-3,
                                   atol=1e-4 * gn + 1.5e-5 + 3e-3 * float(np.abs(fx["grad_sample/" + k]).max()), err_msg=k)
    if chaotic_prefix:
        print(f"\ntrain step vs the reference: loss rel {abs(loss - float(fx['train_loss'])) / abs(loss):.2e}, worst gradient-norm deviation "
              f"{worst_norm:.2e}")
    sd = model.state_dict()
    for key, ref in fx.items():
        if key.startswith("bn_after/"):
            mine = sd[key[len("bn_after/"):]]
            mine = mine.cpu().numpy() if key.endswith("num_batches_tracked") else sample_flat(mine, 16)
            if chaotic_prefix and key[len("bn_after/"):].startswith(chaotic_prefix) and not key.endswith("num_batches_tracked"):
                # (batch statistics over a point set that one flipped sample changes: 1e-5 absolute was observed)
                np.testing.assert_allclose(mine, ref, rtol=chaotic_bn_rtol, atol=1e-4, err_msg=key)
                continue
            np.testing.assert_allclose(mine, ref, rtol=2e-4, atol=2e-6, err_msg=key)
    for k, p in model.named_parameters():
        if p.grad is None or (chaotic_prefix and k.startswith(chaotic_prefix)):
            continue
        g = fx["grad_sample/" + k]
        # Adam's first step is -lr*sign(g) up to eps: only entries whose sign is not at the noise floor
        sel = np.abs(g) > max(1e-4, 0.02 * float(np.abs(g).max()))
        np.testing.assert_allclose(sample_flat(p.detach() - before[k], 16)[sel], fx["delta_sample/" + k][sel],
                                   rtol=2e-3, atol=1e-7, err_msg=k)
    return loss


@pytest.mark.parametrize("mtype", ["forward", "backward", "arbitrary"])
def test_train_step_matches_golden(mtype):
    fx, cfg, seed, data = fixture_setup("tiny_" + mtype, mtype)
    model, train_fn, _ = build_product(cfg, seed, DEV)
    _check_train_step(fx, model, train_fn, cfg, data)


@pytest.mark.parametrize("mtype", ["forward", "arbitrary"])
def test_eight_train_steps_track_the_oracle(mtype):
    """Several optimizer steps in a row (weight packs rebuilt after each Adam step, BatchNorm statistics carried along):
    the loss curve of the HIP path against the CPU oracle's on the same inputs, both started from the same weights.  One step
    is checked in detail against the reference's fixtures above; this one catches state that goes stale BETWEEN steps."""
    from nsdp_amd.model import optimizer_factory
    npl = [256, 64, 16]
    cfg = model_cfg(mtype, npl)
    data = synth.make_batch(77, 2, 256, 128)
    model, train_fn, state = build_product(cfg, 77, DEV)
    model.train()
    lr = 5e-5       # (at the config's 5e-4 the loss of these untrained weights oscillates 0.22 -> 0.09 -> 0.28 -> 0.07: a chaotic
    #                  trajectory amplifies rounding differences to percents within four steps, in the oracle itself too)
    _, opt = optimizer_factory({"optimizer": "Adam", "lr": lr}, model.parameters())
    dd = to_dev(data, DEV)
    snap = snapshot_model(model)
    got = [train_fn(model, opt, dd, cfg) for _ in range(8)]
    sd = tdnet_ref.to_torch_state(state, requires_grad=True)
    ref_opt = torch.optim.Adam([sd[k] for k in tdnet_ref.trainable(sd)], lr=lr)
    cpu = {k: torch.from_numpy(v) for k, v in data.items()}
    ref = [tdnet_ref.train_step(sd, cfg["model"], cpu, ref_opt) for _ in range(8)]
    print(f"\n{mtype}: HIP {[round(v, 5) for v in got]}\n{' ' * len(mtype)}  CPU {[round(v, 5) for v in ref]}")
    assert ref[-1] < ref[0]                                   # (it does train)
    if mtype == "forward":
        for a, b in zip(got, ref):
            assert abs(a - b) <= 5e-3 * abs(b) + 1e-6, (got, ref)
    else:
        # the second network samples and groups the first network's PREDICTED points: one flipped neighbour after an update
        # and the two curves part (0.188 against 0.164 at the third step) -- only the first two steps are pinned.  The second
        # step already sits inside that band: four numerically equivalent variants of this library (NSDP_BN_SLAB x
        # NSDP_ENCODE_ONCE) give 0.2111 / 0.2111 / 0.2043 / 0.2081 against the oracle's 0.2051
        assert abs(got[0] - ref[0]) <= 1e-4 * ref[0] and abs(got[1] - ref[1]) <= 4e-2 * ref[1], (got, ref)
        assert got[-1] < 0.7 * got[0] and ref[-1] < 0.7 * ref[0], (got, ref)
        # ... and so that the 4 % above cannot hide a real regression: the SAME library with the batch mean summed in another
        # order (three-launch BatchNorm) is the variant that lands next to the oracle, and it is held to the original 2 %
        restore_model(model, snap)
        _, opt2 = optimizer_factory({"optimizer": "Adam", "lr": lr}, model.parameters())
        with batchnorm_three_launch():
            alt = [train_fn(model, opt2, dd, cfg) for _ in range(2)]
        print(f"{' ' * len(mtype)}  HIP, three-launch BatchNorm {[round(v, 5) for v in alt]}")
        assert abs(alt[0] - ref[0]) <= 1e-4 * ref[0] and abs(alt[1] - ref[1]) <= 2e-2 * ref[1], (alt, ref)


def test_full_shape_train_step_matches_golden():
    """BASELINE configs[0] exactly: forward.yaml, B = 1, 2048 surface + 8192 query points, one train step -- against the
    imported reference's loss / gradients / BN statistics / Adam deltas (tests/golden/full_forward.npz).  At this size
    the decoder's dense layers (57 344 rows) and the first encoder block (20 480 .. 32 000 rows) are on the bf16x3
    kernels (forward, dX and weight gradients)."""
    fx, cfg, seed, data = fixture_setup("full_forward", "forward")
    model, train_fn, _ = build_product(cfg, seed, DEV)
    with _variant_trace() as names:
        _check_train_step(fx, model, train_fn, cfg, data)
    assert any(n.startswith("linear_bf16x3<") for n in names), names
    assert any(n.startswith("wgrad_bf16x3<13,13") for n in names), names


def test_full_shape_arbitrary_eval_and_train_step_match_golden():
    """BASELINE config 3's function at its full point counts, against the REFERENCE: arbitrary.yaml (FlowArbitrary,
    reference model/flow_arbitrary.py:15-48 -- canonicalise with the 'backward' TDNet, deform with the 'forward' one) at
    B = 2, 2048 surface + 8192 query points, fp32.  Expected values: the imported reference run on CPU
    (tests/golden/full_arbitrary.npz, oracle/make_golden.py --arbitrary-full): eval output, train-step loss, every
    gradient norm + samples, None-gradient set, BatchNorm statistics, Adam deltas."""
    fx, cfg, seed, data = fixture_setup("full_arbitrary", "arbitrary")
    assert (int(fx["meta_batch"]), int(fx["meta_ns"]), int(fx["meta_nq"])) == (2, 2048, 8192)
    model, train_fn, _ = build_product(cfg, seed, DEV)
    model.eval()
    with torch.no_grad():
        out = run_forward(model, cfg, to_dev(data, DEV)).cpu().numpy()
    s = int(fx["meta_eval_stride"])
    err = np.sqrt(((out[:, ::s].astype(np.float64) - fx["eval_out"]) ** 2).sum(-1))
    # (unstable-argsort ties of the reference's kNN, see test_full_shape_forward_matches_golden: at most a couple of queries)
    l2 = float(np.sqrt((np.sort(err, axis=1)[:, :-2] ** 2).mean(-1)).max())
    print(f"\nFlowArbitrary full size, eval L2 vs the reference: {l2:.2e}")
    assert l2 <= TOL_L2, l2
    with _variant_trace() as names:
        # (both networks: the second one samples, groups and queries at the first one's PREDICTED points, so every parameter
        # of FlowArbitrary sits upstream of some discrete selection; on one box this step met the tight bars of
        # _check_train_step throughout, on the next a 0.5 % / 2 % deviation appeared in two encoder gradients -- loss,
        # eval output and the None-gradient set are held tight, gradient norms to 5 %, BatchNorm statistics to 2 %; the tiny
        # FlowArbitrary fixture stays on the tight bars)
        _check_train_step(fx, model, train_fn, cfg, data, chaotic_prefix="model_")
    assert any(n.startswith("linear_bf16x3<") for n in names), names


def test_full_shape_arbitrary_replayed_train_step_matches_golden():
    """The same FlowArbitrary step as above (B = 2, 2048 + 8192 points, the imported reference's fixture) as bench.py's config-3
    launcher runs it: captured once, REPLAYED through the multi-stream executor -- loss, None-gradient set, gradient norms,
    BatchNorm statistics of the replay against the reference (the replay-equals-eager race detector for this model:
    tests/test_graph_exec_gpu.py::test_replayed_arbitrary_and_bf16_steps_are_bit_equal_to_the_eager_step)."""
    fx, cfg, seed, data = fixture_setup("full_arbitrary", "arbitrary")
    model, train_fn, _ = build_product(cfg, seed, DEV)
    _check_train_step(fx, model, train_fn, cfg, data, chaotic_prefix="model_", replay=True)


def test_b8_arbitrary_eval_matches_golden_and_train_step_matches_the_oracle_network_by_network():
    """FlowArbitrary (BASELINE config 3's function) at B = 8 full-size shapes: 458 752 rows in each decoder's attention layers and
    163 840 in the first encoder blocks, i.e. the at-scale bf16x3 forward / dX / weight-gradient kernels, the one-hot scatter and
    the side stream -- the code paths of the timed B = 32 step, which full_arbitrary.npz (B = 2) only partly takes.

    FlowArbitrary is a COMPOSITION in which network 2 samples (FPS) and groups (kNN) the points network 1 PREDICTS, so a 1e-6
    difference in those points flips a farthest-point choice -- and every later choice of that shape -- in most shapes.  The
    reference does this to itself: the same reference model on the same inputs with 3 instead of 8 CPU threads differs by
    1.3e-6 (median) in one shape's canonical points and by 5.1e-4 L2 in that shape's output.  Measured for this library at this
    size: network 1 agrees with the oracle to 3.9e-6 (max abs), network 2 ON THE ORACLE'S canonical points agrees to 6e-7 L2
    per shape (loss 0.19710150 against 0.19710149), the end-to-end output to 9e-5 ... 2.3e-3 per shape and the end-to-end
    train loss to 2.4e-3 -- the composition's conditioning, not a kernel's error.  Hence three checks:
      1. eval mode against the REFERENCE (tests/golden/b8_arbitrary.npz, oracle/make_golden.py --arbitrary-b8): every tap of
         network 1 at the 2e-4 bar; the final output inside the composition's chaos band per shape;
      2. train mode against the ORACLE (pinned to the reference, run here on the host), network by network: network 1's
         predicted points <= 2e-5, network 2 fed the oracle's points <= 1e-5 L2 per shape, loss to 2e-5;
      3. the GRADIENTS of that train step with network 2's inputs pinned to the oracle's VALUES (the gradient still flows
         through this library's network 1): every parameter's gradient norm against the oracle's autograd at 3e-3 -- the bar
         of the forward-model fixtures, with both networks, the coordinate gradients between them and the encode-once
         decoder pass in the graph."""
    fx, cfg, seed, data = fixture_setup("b8_arbitrary", "arbitrary")
    assert (int(fx["meta_batch"]), int(fx["meta_ns"]), int(fx["meta_nq"])) == (8, 2048, 8192)
    model, train_fn, state = build_product(cfg, seed, DEV)
    dd = to_dev(data, DEV)
    # ---- 1. eval mode against the reference's fixture ------------------------------------------------------------------
    model.eval()
    from nsdp_amd import hip_decoder
    tape = {}
    hs = _hooks(model, tape)
    hip_decoder.ENABLED = False                  # (layer-by-layer decoders: expose the taps)
    try:
        with torch.no_grad():
            out = run_forward(model, cfg, dd).cpu().numpy()
    finally:
        hip_decoder.ENABLED = True
        for h in hs:
            h.remove()
    checked = 0
    for key, ref in fx.items():
        if key.startswith("eval_tap/model_canonicalize."):
            mine = tape[key[len("eval_tap/"):]]
            if key.startswith("eval_tap/model_canonicalize.decoder") and mine.shape[1] != int(fx["meta_ns"]):
                mine = mine[:, int(fx["meta_nq"]):].contiguous()       # (see test_eval_forward_matches_golden)
            np.testing.assert_allclose(sample_flat(mine, 64), ref, rtol=0, atol=2e-4, err_msg=key)
            checked += 1
    assert checked >= 15, checked
    s = int(fx["meta_eval_stride"])
    err = np.sqrt(((out[:, ::s].astype(np.float64) - fx["eval_out"]) ** 2).sum(-1))
    per_shape = np.sqrt((np.sort(err, axis=1)[:, :-2] ** 2).mean(-1))
    print(f"\nFlowArbitrary B = 8 full size, eval L2 vs the reference per shape: {np.array2string(per_shape, precision=2)}")
    assert np.median(per_shape) <= 3e-3 and per_shape.max() <= 2e-2, per_shape
    del tape
    # ---- 2. train mode, network by network, against the oracle ------------------------------------------------------------
    torch.set_num_threads(min(16, torch.get_num_threads() or 1) or 1)
    sd = tdnet_ref.to_torch_state(state, requires_grad=True)
    cpu = {k: torch.from_numpy(v) for k, v in data.items()}
    inputs = cpu["surface_samples_inputs"]
    src, tgt, mask = inputs[:, :, 0:3], inputs[:, :, 3:6], inputs[:, :, 6:7]
    mc = cfg["model"]
    q2c = tdnet_ref.deformation_network(sd, mc, cpu["space_samples_src"], src, True, True, "model_canonicalize.", None)
    s2c = tdnet_ref.deformation_network(sd, mc, src, src, True, True, "model_canonicalize.", None)
    ref = tdnet_ref.deformation_network(sd, mc, q2c, torch.cat([s2c, tgt, mask], dim=-1).contiguous(), False, True,
                                        "model_deform.", None)
    ref_loss = tdnet_ref.compute_l2_error(ref, cpu["space_samples_tgt"])
    ref_loss.backward()
    model.train()
    q2c_d, s2c_d = q2c.detach().to(DEV), s2c.detach().to(DEV)
    with _variant_trace() as names:
        orig = model.canonicalize
        seen = {}

        def pinned(query_sets, surface):
            a, b = orig(query_sets, surface)
            seen["a"], seen["b"] = a.detach(), b.detach()
            # the oracle's VALUES into network 2 (identical inputs -> identical FPS / kNN decisions), this library's GRADIENT path
            return [q2c_d + (a - a.detach()), s2c_d + (b - b.detach())]
        model.canonicalize = pinned
        try:
            loss = train_fn.loss_fn(model, dd, cfg)
            loss.backward()
        finally:
            model.canonicalize = orig
    d1 = max(float((seen["a"] - q2c_d).abs().max()), float((seen["b"] - s2c_d).abs().max()))
    print(f"train mode: network 1's predicted points vs the oracle {d1:.2e} (max abs); loss {float(loss):.8f} vs {float(ref_loss):.8f}")
    assert d1 <= 2e-5, d1
    assert abs(float(loss) - float(ref_loss)) <= 2e-5 * float(ref_loss), (float(loss), float(ref_loss))
    # ---- 3. gradients of both networks against the oracle's autograd -------------------------------------------------------
    none = sorted(k for k, p in model.named_parameters() if p.grad is None)
    assert none == sorted(str(x) for x in fx["none_grads"])
    worst = (0.0, None)
    for k, p in model.named_parameters():
        if p.grad is None:
            continue
        gn = float(sd[k].grad.double().norm())
        mine = float(p.grad.double().norm())
        dev = abs(mine - gn) / max(gn, 1e-30)
        if gn > 1e-4 and dev > worst[0]:
            worst = (dev, k)
        assert abs(mine - gn) <= 3e-3 * gn + 2e-5, (k, mine, gn)
        sg = sample_flat(sd[k].grad, 16)
        # entries: 16 samples per tensor at 1.5 % of the largest sampled entry (sums with heavy cancellation, e.g. BatchNorm
        # weights of network 1, whose gradient arrives through ~70 layers and the coordinate path of network 2), ONE outlier
        # allowed: a channel that is active in a handful of the 800 rows of the 100-anchor level loses or gains a third of its
        # gradient when one pre-activation at 1e-7 changes sign (observed: elementwise_extras.1.conv2.weight[204, 204], the
        # other 15 samples equal to three digits).  The norms carry the bar.
        mg = sample_flat(p.grad, 16)
        bad = np.abs(mg - sg) > 5e-3 * np.abs(sg) + 1e-4 * gn + 1.5e-5 + 1.5e-2 * float(np.abs(sg).max())
        assert int(bad.sum()) <= 1, (k, mg, sg)
    print(f"worst gradient-norm deviation from the oracle (norms > 1e-4): {worst[0]:.2e} ({worst[1]})")
    assert any(n.startswith("linear_bf16x3<") and n.split(",")[3] == "8" for n in names), names
    assert any(n.startswith("wgrad_bf16x3<13,13") for n in names), names


def test_b16_train_step_matches_golden():
    """The code path bench.py times, under the reference: B = 16 shapes of 2048 / 8192 points (131 072 rows at the
    output layer -> weight gradients on the side stream; 917 504 rows in the decoder's attention layers -> the 8-wave
    bf16x3 GEMM, the LDS-table attention backward and the register-table scatter; 327 680 rows in the first encoder
    block).  Expected values: the imported reference run on CPU at the same size (tests/golden/b16_forward.npz)."""
    from nsdp_amd import hip_linear
    fx, cfg, seed, data = fixture_setup("b16_forward", "forward")
    assert int(fx["meta_batch"]) * int(fx["meta_nq"]) >= hip_linear._OVERLAP_MIN_ROWS
    model, train_fn, _ = build_product(cfg, seed, DEV)
    model.eval()
    with torch.no_grad():
        out = run_forward(model, cfg, to_dev(data, DEV)).cpu().numpy()
    s = int(fx["meta_eval_stride"])
    # (the handful of queries whose 7th/8th anchor distances tie bit-for-bit -- unstable argsort in the reference --
    # cannot move this norm above the bar: 1 of 8192 in the B = 1 fixture)
    err = np.sqrt(((out[:, ::s].astype(np.float64) - fx["eval_out"]) ** 2).sum(-1))
    assert float(np.sqrt((np.sort(err, axis=1)[:, :-2] ** 2).mean(-1)).max()) <= TOL_L2
    used_side = []
    orig = hip_linear._wgrad_deferred
    hip_linear._wgrad_deferred = lambda *a, **k: (used_side.append(1), orig(*a, **k))[1]
    try:
        with _variant_trace() as names:
            _check_train_step(fx, model, train_fn, cfg, data)
    finally:
        hip_linear._wgrad_deferred = orig
    if hip_linear._OVERLAP_WGRAD == "auto" and hip_linear._PARAM_GRADS_DIRECT:     # (NSDP_PARAM_GRADS=autograd: main stream)
        assert len(used_side) > 50, "weight gradients did not take the side stream"
    eight_wave = [n for n in names if n.startswith("linear_bf16x3<") and n.split(",")[3] == "8"]
    assert eight_wave, names
    from nsdp_amd.model import ops
    masked = () if ops.PAIR_MASK else ("wgrad_bf16x3<13,13,mask,notail>",)   # (NSDP_PAIR_MASK=1 removes the masked variants)
    g16 = hip_linear.G16 and not ops.PAIR_MASK
    if g16:      # the hidden tensors of fc_gamma (and their gradients) in the G16 layout: every form of the pair ran
        masked = ("wgrad_bf16x3<13,13,bits,notail> g16:dy" if hip_linear.G16_BITS else "wgrad_bf16x3<13,13,mask,notail> g16:dy",)
        assert any(n.startswith("linear_bf16x3<") and n.endswith(" g16:y") for n in names), sorted(names)
        assert any(n.startswith("linear_bf16x3<") and n.endswith(" g16:x") for n in names), sorted(names)
    from nsdp_amd import hip_attention
    # (the decoder's anchor-table gradients: scatter as a GEMM + the atomics-free attention backward, or the fp32-atomic
    # kernels under NSDP_ONEHOT_SCATTER_F32=0)
    scatter = (("attn_post_bwd_det", "scatter_rows_onehot_f32<8,13,notail>") if hip_attention.ONEHOT_SCATTER_F32
               else ("attn_post_bwd_lds", "scatter_rows_regtab<8>"))
    plain = "wgrad_bf16x3<13,13,plain,notail> g16:x" if g16 else "wgrad_bf16x3<13,13,plain,notail>"
    for needed in scatter + (plain,) + masked:
        assert needed in names, (needed, sorted(names))


def test_b16_replayed_train_step_matches_golden():
    """The launcher bench.py times, under the reference: the SAME B = 16 full-size step as above, but captured and replayed
    by the multi-stream graph executor (csrc/graph_exec.hip) -- loss, every gradient, BatchNorm statistics and Adam deltas
    of the replayed step against the imported reference's (tests/golden/b16_forward.npz)."""
    fx, cfg, seed, data = fixture_setup("b16_forward", "forward")
    model, train_fn, _ = build_product(cfg, seed, DEV)
    _check_train_step(fx, model, train_fn, cfg, data, replay=True)


def test_b32_headline_shard_matches_golden_eager_and_replayed():
    """The headline's own per-GPU shard -- B = 32 shapes of 2048 surface + 8192 query points, exactly what bench.py times --
    under the reference: eval output, then one train step run EAGERLY and one run as a REPLAY of the captured step (the
    default launcher of bench.py), each against the imported reference's loss / gradients / BatchNorm statistics / Adam
    deltas at the same size (tests/golden/b32_forward.npz, oracle/make_golden.py --b32: ~30 GB of CPU temporaries)."""
    fx, cfg, seed, data = fixture_setup("b32_forward", "forward")
    assert (int(fx["meta_batch"]), int(fx["meta_ns"]), int(fx["meta_nq"])) == (32, 2048, 8192)
    model, train_fn, _ = build_product(cfg, seed, DEV)
    model.eval()
    with torch.no_grad():
        out = run_forward(model, cfg, to_dev(data, DEV)).cpu().numpy()
    s = int(fx["meta_eval_stride"])
    err = np.sqrt(((out[:, ::s].astype(np.float64) - fx["eval_out"]) ** 2).sum(-1))
    l2 = float(np.sqrt((np.sort(err, axis=1)[:, :-2] ** 2).mean(-1)).max())
    print(f"\nB = 32 full size, eval L2 vs the reference: {l2:.2e}")
    assert l2 <= TOL_L2, l2
    _check_train_step(fx, model, train_fn, cfg, data)
    del model
    torch.cuda.empty_cache()
    model, train_fn, _ = build_product(cfg, seed, DEV)
    _check_train_step(fx, model, train_fn, cfg, data, replay=True)


def test_b8_full_shape_eval_matches_oracle():
    """BASELINE config 2: forward TDNet, batch = 8 synthetic shapes, 2048 / 8192 points, fp32 -- product (fused decoder
    kernel AND the layer-by-layer path) vs the CPU oracle on the same seeded inputs, <= 1e-4 L2."""
    from nsdp_amd import hip_decoder
    cfg = model_cfg("forward", [2048, 500, 100])
    data = synth.make_batch(808, 8, 2048, 8192)
    model, _, state = build_product(cfg, 808, DEV)
    model.eval()
    torch.set_num_threads(min(16, torch.get_num_threads() or 1) or 1)
    with torch.no_grad():
        sd = tdnet_ref.to_torch_state(state)
        ref = tdnet_ref.model_forward(sd, cfg["model"], {k: torch.from_numpy(v) for k, v in data.items()}).numpy()
        out = run_forward(model, cfg, to_dev(data, DEV)).cpu().numpy()
        hip_decoder.ENABLED = False
        try:
            out_layered = run_forward(model, cfg, to_dev(data, DEV)).cpu().numpy()
        finally:
            hip_decoder.ENABLED = True
    assert l2_err(out, ref) <= TOL_L2
    assert l2_err(out_layered, ref) <= TOL_L2


@pytest.mark.parametrize("mtype,seed", [("forward", 5), ("backward", 6)])
def test_eval_forward_matches_oracle_other_inputs(mtype, seed):
    """Same seeded inputs through the HIP path and the CPU oracle (ragged sizes, batch 3)."""
    npl = [300, 70, 20]
    cfg = model_cfg(mtype, npl)
    data = synth.make_batch(seed, 3, 300, 203)
    model, _, state = build_product(cfg, seed, DEV)
    model.eval()
    with torch.no_grad():
        out = run_forward(model, cfg, to_dev(data, DEV)).cpu().numpy()
        sd = tdnet_ref.to_torch_state(state)
        ref = tdnet_ref.model_forward(sd, cfg["model"], {k: torch.from_numpy(v) for k, v in data.items()}).numpy()
    assert l2_err(out, ref) <= TOL_L2


def test_dense_inference_step_matches_oracle():
    """test_on_batch_with_cano (reference deformation_networks.py:90-109): surface samples, then all mesh
    vertices through the same encoder input -- the dense-inference call pattern of BASELINE config 5."""
    from nsdp_amd.model import build_model
    cfg = model_cfg("forward", [256, 64, 16])
    data = synth.make_batch(31, 1, 256, 5)
    model, _, state = build_product(cfg, 31, DEV)
    _, _, _, test_fn = build_model(cfg, device="cpu")
    model.eval()
    verts = synth.uniform(31, "verts", (1, 3001, 3), -0.5, 0.5)
    dd = to_dev(data, DEV)
    dd["surface_samples_src"] = dd["surface_samples_inputs"][:, :, :3].contiguous()
    dd["verts_src"] = torch.from_numpy(verts).to(DEV)
    dd["verts_tgt"] = dd["verts_src"]
    loss, out = test_fn(model, dd, cfg, compute_loss=True)
    sd = tdnet_ref.to_torch_state(state)
    with torch.no_grad():
        ref_v = tdnet_ref.model_forward(sd, cfg["model"], {"surface_samples_inputs": torch.from_numpy(data["surface_samples_inputs"]),
                                                            "q": torch.from_numpy(verts)}, queries_key="q").numpy()
        ref_s = tdnet_ref.model_forward(sd, cfg["model"], {"surface_samples_inputs": torch.from_numpy(data["surface_samples_inputs"]),
                                                            "q": torch.from_numpy(data["surface_samples_inputs"][:, :, :3].copy())},
                                        queries_key="q").numpy()
    assert l2_err(out["verts_tgt_pred"].cpu().numpy(), ref_v) <= TOL_L2
    assert l2_err(out["surface_samples_tgt_pred"].cpu().numpy(), ref_s) <= TOL_L2
    assert abs(loss - float(((ref_v - verts) ** 2).sum(-1).mean() / 2)) < 1e-5


def test_config5_dense_inference_b4_100k_queries_matches_oracle():
    """BASELINE config 5 at its full size: 4 shapes x 100 000 query points through the fused decoder kernel.  A query's
    output depends on the encoder's latent and on that query alone, so the CPU oracle is run on a 3000-query sample per shape
    (both ends of the 100 k, where a ragged last tile would show) and compared with the same rows of the full run; the
    layer-by-layer path is compared on all 400 k rows."""
    from nsdp_amd import hip_decoder
    cfg = model_cfg("forward", [2048, 500, 100])
    nq = 100000
    data = synth.make_batch(555, 4, 2048, nq)
    model, _, state = build_product(cfg, 555, DEV)
    model.eval()
    dd = to_dev(data, DEV)
    with torch.no_grad():
        out = run_forward(model, cfg, dd)
        hip_decoder.ENABLED = False
        try:
            out_layered = run_forward(model, cfg, dd)
        finally:
            hip_decoder.ENABLED = True
        pick = np.concatenate([np.arange(0, 1500), np.arange(nq - 1500, nq)])
        q = data["space_samples_src"][:, pick].copy()
        sd = tdnet_ref.to_torch_state(state)
        torch.set_num_threads(min(16, torch.get_num_threads() or 1) or 1)
        ref = tdnet_ref.model_forward(sd, cfg["model"], {"surface_samples_inputs": torch.from_numpy(data["surface_samples_inputs"]),
                                                          "q": torch.from_numpy(q)}, queries_key="q").numpy()
    assert out.shape == (4, nq, 3)
    assert l2_err(out[:, pick].cpu().numpy(), ref) <= TOL_L2
    assert l2_err(out_layered[:, pick].cpu().numpy(), ref) <= TOL_L2
    assert l2_err(out.cpu().numpy(), out_layered.cpu().numpy()) <= TOL_L2


def _with_encode_once(flag, fn):
    from nsdp_amd.model import deformation_networks as dn
    prev, dn.ENCODE_ONCE = dn.ENCODE_ONCE, flag
    try:
        return fn()
    finally:
        dn.ENCODE_ONCE = prev


def test_encode_once_train_step_equals_the_reference_op_sequence():
    """FlowArbitrary with ONE canonicalise-encoder pass per step (the default) against the reference's op sequence
    (model/flow_arbitrary.py:19-20: two passes over the same cloud; NSDP_ENCODE_ONCE=0 here): same loss, same gradients
    (two decoder paths summed into one encoder backward instead of two backward passes), same BatchNorm buffers including
    num_batches_tracked == 2, same weights after the Adam step.  Both forms are also held to the reference's fixture by
    test_train_step_matches_golden / test_full_shape_arbitrary_*; this one pins them against each other at tight bars."""
    from nsdp_amd.model import optimizer_factory
    fx, cfg, seed, data = fixture_setup("tiny_arbitrary", "arbitrary")
    dd = to_dev(data, DEV)
    res = {}
    for flag in (True, False):
        def run():
            model, train_fn, _ = build_product(cfg, seed, DEV)
            model.train()
            _, opt = optimizer_factory({"optimizer": "Adam", "lr": 5e-4}, model.parameters())
            loss = train_fn(model, opt, dd, cfg)
            return loss, {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}, \
                {k: v.clone() for k, v in model.state_dict().items()}
        res[flag] = _with_encode_once(flag, run)
    (l1, g1, s1), (l0, g0, s0) = res[True], res[False]
    assert abs(l1 - l0) <= 1e-6 * abs(l0), (l1, l0)
    assert sorted(g1) == sorted(g0)
    for k in g0:
        # (absolute floor: gradients that are analytically zero -- a bias in front of a train-mode BatchNorm, e.g.
        # transformer_begin.fc_delta.2.bias -- are fp32 cancellation noise of ~1e-3 whose value follows the summation order)
        # Relative bars, not bit bars: the decoder over the concatenated queries takes other tile shapes than two separate
        # calls, network 1's output points differ in the last bit (2.7e-7), and network 2 samples / groups at those points --
        # one neighbour that flips moves single gradient entries by a percent (a dropped or doubled path would move norms by
        # tens of percent)
        d = (g1[k] - g0[k]).double()
        assert float(d.abs().max()) <= 2e-2 * float(g0[k].abs().max()) + 3e-6, k
        assert float(d.norm()) <= 1e-2 * float(g0[k].double().norm()) + 3e-6, k
    nbt = [k for k in s0 if k.endswith("num_batches_tracked")]
    assert any(int(s0[k]) == 2 for k in nbt) and any(int(s0[k]) == 1 for k in nbt)
    for k in s0:
        if k.endswith("num_batches_tracked"):
            assert int(s1[k]) == int(s0[k]), k
        elif "running_" in k:
            assert float((s1[k] - s0[k]).abs().max()) <= 1e-6 * (1.0 + float(s0[k].abs().max())), k


@pytest.mark.parametrize("mtype", ["forward", "arbitrary"])
def test_encode_once_dense_inference_equals_the_reference_op_sequence(mtype):
    """test_on_batch_with_cano / _with_arbitrary: one encoder pass per distinct surface cloud (1 instead of 2, 2 instead of 6)
    against the reference's sequence of whole-module calls (NSDP_ENCODE_ONCE=0)."""
    from nsdp_amd.model import build_model
    cfg = model_cfg(mtype, [256, 64, 16])
    data = synth.make_batch(41, 2, 256, 5)
    model, _, _ = build_product(cfg, 41, DEV)
    _, _, _, test_fn = build_model(cfg, device="cpu")
    model.eval()
    verts = synth.uniform(41, "verts", (2, 1501, 3), -0.5, 0.5)
    outs = {}
    for flag in (True, False):
        dd = to_dev(data, DEV)
        dd["surface_samples_src"] = dd["surface_samples_inputs"][:, :, :3].contiguous()
        dd["verts_src"] = torch.from_numpy(verts).to(DEV)
        dd["verts_tgt"] = dd["verts_src"]
        loss, out = _with_encode_once(flag, lambda: test_fn(model, dd, cfg, compute_loss=True))
        outs[flag] = (loss, out["surface_samples_tgt_pred"].clone(), out["verts_tgt_pred"].clone())
    assert abs(outs[True][0] - outs[False][0]) <= 1e-6
    for a, b in zip(outs[True][1:], outs[False][1:]):
        assert a.shape == b.shape
        assert l2_err(a.cpu().numpy(), b.cpu().numpy()) <= 1e-6


def test_flat_bucket_gradients_equal_plain_gradients_on_gpu():
    """GradAllReducer's .grad views (the data-parallel path) must receive exactly what plain autograd
    produces, including through the hand-written backward kernels."""
    from nsdp_amd.parallel import GradAllReducer
    from nsdp_amd.model.utils import compute_l2_error
    cfg = model_cfg("forward", [256, 64, 16])
    data = to_dev(synth.make_batch(8, 2, 256, 128), DEV)
    model, _, _ = build_product(cfg, 8, DEV)
    model.train()
    torch.manual_seed(0)
    compute_l2_error(model(data["space_samples_src"], data["surface_samples_inputs"]), data["space_samples_tgt"]).backward()
    plain = {k: p.grad.clone() for k, p in model.named_parameters()}
    model2, _, _ = build_product(cfg, 8, DEV)
    model2.train()
    red = GradAllReducer(model2, 1)
    red.zero_grad()
    compute_l2_error(model2(data["space_samples_src"], data["surface_samples_inputs"]), data["space_samples_tgt"]).backward()
    red.all_reduce_mean()
    for k, p in model2.named_parameters():
        assert p.grad.data_ptr() == dict(zip([n for n, _ in red.named], red.views))[k].data_ptr()
        g = plain[k]
        if nondeterministic_knobs():       # (the A/B forms with floating-point atomics: ordering noise in the scatter-adds)
            assert float((p.grad - g).abs().max()) <= 2e-3 * float(g.abs().max()) + 1e-6, k
        else:                              # the step is bit-reproducible: a view into the flat bucket receives the SAME bits
            assert torch.equal(p.grad, g), k


@pytest.mark.parametrize("mtype", ["forward", "arbitrary"])
def test_forward_after_optimizer_steps_uses_the_new_weights(mtype):
    """Weight packs (per-layer and the fused decoder's) are caches of the parameters: after optimizer steps -- here
    with PyTorch's fused Adam, which does NOT bump tensor version counters -- the model must compute exactly what a
    freshly built model holding the same state_dict computes."""
    fx, cfg, seed, data = fixture_setup("tiny_" + mtype, mtype)
    model, train_fn, _ = build_product(cfg, seed, DEV)
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=5e-3, fused=True)
    dev_data = to_dev(data, DEV)
    l0 = train_fn(model, opt, dev_data, cfg)
    l1 = train_fn(model, opt, dev_data, cfg)
    l2 = train_fn(model, opt, dev_data, cfg)
    assert len({round(l0, 9), round(l1, 9), round(l2, 9)}) == 3          # the loss moves with the weights
    fresh, _, _ = build_product(cfg, seed, DEV)
    fresh.load_state_dict(model.state_dict())
    for m in (model, fresh):
        m.eval()
    with torch.no_grad():
        a = run_forward(model, cfg, dev_data)                             # fused decoder kernel (its own pack)
        b = run_forward(fresh, cfg, dev_data)
    assert torch.equal(a, b)
    for m in (model, fresh):
        m.train()
    a = run_forward(model, cfg, dev_data)                                 # layer-by-layer path, packs per parameter
    b = run_forward(fresh, cfg, dev_data)
    assert torch.equal(a, b)
