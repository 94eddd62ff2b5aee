"""fp32-MFMA dense layer kernels vs a plain PyTorch fp32 reference of the same op (forward + backward)."""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _ref(x, w, b, relu_in, relu_out, residual):
    xi = F.relu(x) if relu_in else x
    y = F.linear(xi.double(), w.double(), None if b is None else b.double())
    if residual is not None:
        y = y + residual.double()
    return F.relu(y) if relu_out else y


CASES = [  # (M, K, N, bias, relu_in, relu_out, residual)
    (1, 4, 16, True, False, False, False),
    (37, 120, 120, True, False, True, False),
    (1000, 128, 256, False, False, False, False),
    (513, 200, 200, True, False, True, False),
    (777, 200, 128, True, False, False, True),
    (300, 128, 128, True, True, True, False),
    (300, 128, 3, True, True, False, False),
    (2048, 3, 120, True, False, True, False),
    (129, 256, 256, True, False, False, True),
    (5000, 4, 120, True, False, False, False),
    (64, 256, 200, False, False, False, False),
    (4099, 120, 256, True, False, False, False),
]


@pytest.mark.parametrize("M,K,N,bias,relu_in,relu_out,res", CASES)
def test_linear_forward_backward(M, K, N, bias, relu_in, relu_out, res):
    from nsdp_amd.hip_linear import linear
    g = torch.Generator(device="cpu").manual_seed(M * 131 + K * 7 + N)
    x = torch.randn(M, K, generator=g).to(DEV).requires_grad_(True)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV).requires_grad_(True)
    b = torch.randn(N, generator=g).to(DEV).requires_grad_(True) if bias else None
    r = torch.randn(M, N, generator=g).to(DEV).requires_grad_(True) if res else None
    y = linear(x, w, b, relu_in=relu_in, relu_out=relu_out, residual=r)
    yr = _ref(x, w, b, relu_in, relu_out, r)
    scale = float(yr.abs().max()) + 1e-6
    assert float((y.double() - yr).abs().max()) <= 2e-6 * scale * max(1.0, K ** 0.5 / 4)
    go = torch.randn(M, N, generator=g).to(DEV)
    grads = torch.autograd.grad(y, [t for t in (x, w, b, r) if t is not None], go)
    grads_ref = torch.autograd.grad(yr, [t for t in (x, w, b, r) if t is not None], go.double())
    for a, e in zip(grads, grads_ref):
        s = float(e.abs().max()) + 1e-6
        assert float((a.double() - e.double()).abs().max()) <= 1e-5 * s, (a.shape, float((a.double() - e).abs().max()), s)


def test_linear_3d_input_and_conv_weight():
    from nsdp_amd.hip_linear import linear
    x = torch.randn(3, 50, 120, device=DEV)
    conv = torch.nn.Conv1d(120, 120, 1).to(DEV)
    y = linear(x, conv.weight, conv.bias)
    ref = conv(x.permute(0, 2, 1)).permute(0, 2, 1)
    assert float((y - ref).abs().max()) < 1e-4


def test_wgrad_is_deterministic():
    from nsdp_amd.hip_linear import _wgrad
    dy = torch.randn(100000, 128, device=DEV)
    x = torch.randn(100000, 200, device=DEV)
    a, da = _wgrad(dy, x, None, False, True)
    b, db = _wgrad(dy, x, None, False, True)
    assert torch.equal(a, b) and torch.equal(da, db)
    ref = dy.double().t() @ x.double()
    assert float((a.double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())


@pytest.mark.parametrize("M,K,N,n_layers", [(4096, 200, 128, 6), (40000, 200, 128, 3), (777, 64, 40, 4)])
def test_input_grad_sum_matches_autograd(M, K, N, n_layers):
    """Fan-out of one input over several layers: the gradient summed inside the dX GEMMs (InputGradSum) equals
    autograd's pairwise sum of the separate dX tensors; weight gradients are untouched; a second backward pass over
    the retained graph works (the link re-arms)."""
    from nsdp_amd import hip_linear
    g = torch.Generator().manual_seed(M + n_layers)
    x0 = torch.randn(M, K, generator=g).to(DEV)
    ws = [(torch.randn(N, K, generator=g) * 0.1).to(DEV) for _ in range(n_layers)]
    go = torch.randn(M, N, generator=g).to(DEV)

    def run(use_link, passes=1):
        x = x0.clone().requires_grad_(True)
        wl = [w.clone().requires_grad_(True) for w in ws]
        link = hip_linear.InputGradSum() if use_link else None
        net = hip_linear.linear(x, wl[0], relu_out=True, grad_sum=link)
        for w in wl[1:]:
            net = hip_linear.linear(x, w, residual=net, grad_sum=link)
        outs = []
        for p in range(passes):
            outs.append(torch.autograd.grad(net, [x] + wl, go, retain_graph=p + 1 < passes))
        return outs

    plain = run(False)[0]
    fused = run(True, passes=2)
    for got in fused:
        for a_, e_ in zip(got, plain):
            assert a_.shape == e_.shape
            assert float((a_ - e_).abs().max()) <= 2e-6 * float(e_.abs().max()) + 1e-6


def test_input_grad_sum_refuses_fused_input_relu():
    from nsdp_amd import hip_linear
    x = torch.randn(64, 32, device=DEV, requires_grad=True)
    w = torch.randn(16, 32, device=DEV, requires_grad=True)
    with pytest.raises(ValueError):
        hip_linear.linear(x, w, relu_in=True, grad_sum=hip_linear.InputGradSum())


def test_failed_backward_leaves_nothing_behind():
    """Weight gradients wait on a side stream until an end-of-backward callback publishes them.  A pass that dies with
    an exception never runs that callback: its leftovers must not leak into the next pass's gradients."""
    from nsdp_amd import hip_linear
    lin1, lin2 = torch.nn.Linear(64, 48).to(DEV), torch.nn.Linear(48, 32).to(DEV)
    x = torch.randn(4096, 64, device=DEV)
    was = hip_linear._OVERLAP_WGRAD
    hip_linear._OVERLAP_WGRAD = True
    try:
        def run(fail):
            for p in (*lin1.parameters(), *lin2.parameters()):
                p.grad = None
            h = hip_linear.linear(x, lin1.weight, lin1.bias, relu_out=True, params=True)
            if fail:
                h.register_hook(lambda g: (_ for _ in ()).throw(RuntimeError("boom")))
            y = hip_linear.linear(h, lin2.weight, lin2.bias, params=True)
            y.square().sum().backward()
            torch.cuda.synchronize()
            return [p.grad.clone() for p in (*lin1.parameters(), *lin2.parameters())]

        good = run(False)
        with pytest.raises(RuntimeError):
            run(True)                      # lin2's weight gradient was already parked on the side stream
        again = run(False)
        for a_, e_ in zip(again, good):
            assert torch.equal(a_, e_)
    finally:
        hip_linear._OVERLAP_WGRAD = was


@pytest.mark.parametrize("M,N,mask,relu_x", [(70001, 200, True, False), (5000, 120, False, False), (131072, 256, True, True),
                                             (33, 16, False, False)])
def test_wgrad_k4_stream_kernel(M, N, mask, relu_x):
    """K = 4 weight gradients (position-encoding layers) take the streaming kernel: against fp64, deterministic."""
    from nsdp_amd import hip_linear
    g = torch.Generator().manual_seed(M + N)
    dy = torch.randn(M, N, generator=g).to(DEV)
    x = torch.randn(M, 4, generator=g).to(DEV)
    m = torch.randn(M, N, generator=g).to(DEV) if mask else None
    dw, db = hip_linear._wgrad(dy, x, m, relu_x, True)
    dyp = dy.double() * (m > 0) if mask else dy.double()
    xp = F.relu(x.double()) if relu_x else x.double()
    ref_w, ref_b = dyp.t() @ xp, dyp.sum(0)
    assert float((dw.double() - ref_w).abs().max()) <= 3e-6 * float(ref_w.abs().max()) + 1e-5
    assert float((db.double() - ref_b).abs().max()) <= 3e-6 * float(ref_b.abs().max()) + 1e-5
    dw2, db2 = hip_linear._wgrad(dy, x, m, relu_x, True)
    assert torch.equal(dw, dw2) and torch.equal(db, db2)


def test_batched_repack_equals_single_packs():
    """After an optimizer step all registered packs are rebuilt by one batched launch: bit-identical to the per-layer
    pack kernels, for both pack formats, forward and transposed."""
    from nsdp_amd import hip_linear
    torch.manual_seed(0)
    layers = [torch.nn.Linear(k, n).to(DEV) for k, n in [(64, 48), (48, 200), (200, 200), (200, 128), (128, 256),
                                                         (256, 120), (120, 64), (64, 16), (16, 40), (40, 8)]]
    x_small = torch.randn(300, 64, device=DEV, requires_grad=True)        # fp32 packs ("wp")
    x_big = torch.randn(40000, 64, device=DEV, requires_grad=True)        # bf16x3 packs where the shapes allow
    opt = torch.optim.SGD([p for l in layers for p in l.parameters()], lr=1e-6)

    def fwd(x):
        for l in layers:
            x = hip_linear.linear(x, l.weight, l.bias, relu_out=True, params=True)
        return x

    for step in range(3):
        opt.zero_grad()
        (fwd(x_small).mean() + fwd(x_big).mean()).backward()
        opt.step()
    fwd(x_small); fwd(x_big)                      # stale caches -> one batched rebuild
    reg = hip_linear._pack_registry[DEV.index]["entries"]
    checked = 0
    for l in layers:
        w = l.weight.detach()
        for kind in ("wp", "x3"):
            ent = reg.get((id(l.weight), kind))
            if ent is None:
                continue
            assert l.weight.__dict__["_nsdp_pack"]["key"] == hip_linear._pack_key(w)
            single = (hip_linear.pack_weight_x3 if kind == "x3" else hip_linear.pack_weight)(w, ent[2] is not None, ent[3] is not None)
            for got, want in zip(ent[2:4], single):
                if got is not None:
                    assert torch.isfinite(w).all()
                    assert torch.equal(got.view(torch.uint8), want.view(torch.uint8))      # bit for bit
                    checked += 1
    assert checked >= 12
    ref = x_small
    for l in layers:
        ref = F.relu(F.linear(ref.double(), l.weight.double(), l.bias.double()))
    out = fwd(x_small)
    assert float((out.double() - ref).abs().max()) <= 1e-5 * (float(ref.abs().max()) + 1.0)


@pytest.mark.parametrize("fused", [False, True])
def test_packs_follow_every_optimizer(fused):
    """torch's fused optimizers update parameters without bumping their version counters: the pack caches must be
    invalidated by the optimizer-step hook, not by versions alone (a stale pack means training on frozen weights)."""
    from nsdp_amd import hip_linear
    torch.manual_seed(1)
    lin = torch.nn.Linear(64, 32).to(DEV)
    x = torch.randn(512, 64, device=DEV)
    opt = torch.optim.Adam(lin.parameters(), lr=0.05, fused=fused)
    for _ in range(3):
        opt.zero_grad()
        y = hip_linear.linear(x, lin.weight, lin.bias, params=True)
        ref = F.linear(x.double(), lin.weight.double(), lin.bias.double())
        assert float((y.double() - ref).abs().max()) <= 1e-5 * (float(ref.abs().max()) + 1.0)
        y.square().mean().backward()
        opt.step()
    with torch.no_grad():
        lin.weight.data.mul_(2.0)                # untracked write: explicit invalidation is the contract
    hip_linear.invalidate_weight_packs()
    y = hip_linear.linear(x, lin.weight, lin.bias, params=True)
    ref = F.linear(x.double(), lin.weight.double(), lin.bias.double())
    assert float((y.double() - ref).abs().max()) <= 1e-5 * (float(ref.abs().max()) + 1.0)


def test_retained_graph_across_an_optimizer_step_fails_loudly():
    """The batched rebuild rewrites pack buffers in place; a graph retained from before the step must not silently
    compute dX with the new weights."""
    from nsdp_amd import hip_linear
    layers = [torch.nn.Linear(32, 32).to(DEV) for _ in range(9)]        # >= _BATCH_MIN registered packs
    opt = torch.optim.SGD([p for l in layers for p in l.parameters()], lr=1e-3)
    x = torch.randn(256, 32, device=DEV, requires_grad=True)

    def fwd():
        h = x
        for l in layers:
            h = hip_linear.linear(h, l.weight, l.bias, relu_out=True, params=True)
        return h.mean()

    fwd().backward()
    opt.step()
    old = fwd()                      # batched rebuild happens here; `old` saves the rebuilt packs
    old.backward(retain_graph=True)
    opt.step()
    fwd()                            # next rebuild rewrites the buffers `old` saved
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        old.backward()


def _layer(seed=0, n=64, k=128):
    torch.manual_seed(seed)
    lin = torch.nn.Linear(k, n).to(DEV)
    x = torch.randn(4096, k, device=DEV, requires_grad=True)
    return lin, x


def test_param_gradients_reach_hooks_and_autograd_grad():
    """Direct publication of weight gradients (params=True) bypasses autograd; anything that observes gradients through
    autograd must still work: a tensor hook / post-accumulate-grad hook switches the layer to the autograd path, and
    `autograd_param_grads()` does the same for torch.autograd.grad / backward(inputs=...)."""
    from nsdp_amd import hip_linear
    lin, x = _layer()
    ref_y = F.linear(x, lin.weight, lin.bias)
    gw_ref, gb_ref = torch.autograd.grad(ref_y.square().sum(), [lin.weight, lin.bias])
    # default: published straight into .grad
    hip_linear.linear(x, lin.weight, lin.bias, params=True).square().sum().backward()
    assert torch.allclose(lin.weight.grad, gw_ref, rtol=1e-4, atol=1e-3)
    lin.zero_grad()
    # tensor hook on the weight: must fire, with the right gradient
    fired = []
    h = lin.weight.register_hook(lambda g: fired.append(g.clone()))
    hip_linear.linear(x, lin.weight, lin.bias, params=True).square().sum().backward()
    h.remove()
    assert len(fired) == 1 and torch.allclose(fired[0], gw_ref, rtol=1e-4, atol=1e-3)
    assert torch.allclose(lin.weight.grad, gw_ref, rtol=1e-4, atol=1e-3)
    assert torch.allclose(lin.bias.grad, gb_ref, rtol=1e-4, atol=1e-3)
    lin.zero_grad()
    # post-accumulate-grad hook (what DDP / FSDP use)
    fired = []
    h = lin.bias.register_post_accumulate_grad_hook(lambda p: fired.append(p.grad.clone()))
    hip_linear.linear(x, lin.weight, lin.bias, params=True).square().sum().backward()
    h.remove()
    assert len(fired) == 1 and torch.allclose(fired[0], gb_ref, rtol=1e-4, atol=1e-3)
    lin.zero_grad()
    # torch.autograd.grad on parameters: inside the context manager it returns them and leaves .grad alone
    with hip_linear.autograd_param_grads():
        y = hip_linear.linear(x, lin.weight, lin.bias, params=True)
        gw, gb = torch.autograd.grad(y.square().sum(), [lin.weight, lin.bias])
    assert torch.allclose(gw, gw_ref, rtol=1e-4, atol=1e-3) and torch.allclose(gb, gb_ref, rtol=1e-4, atol=1e-3)
    assert lin.weight.grad is None and lin.bias.grad is None


def test_no_grad_forward_reuses_the_weight_pack():
    """Eval / no_grad forwards must hit the per-parameter pack cache (one pack kernel per layer per model lifetime, not
    per call) and still see weight updates."""
    from nsdp_amd import hip_linear
    lin, x = _layer(1)
    calls = []
    orig = hip_linear.pack_weight
    hip_linear.pack_weight = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        with torch.no_grad():
            y0 = hip_linear.linear(x, lin.weight, lin.bias, params=True)
            y1 = hip_linear.linear(x, lin.weight, lin.bias, params=True)
            assert len(calls) == 1 and torch.equal(y0, y1)
            lin.weight.mul_(2.0)                         # in-place edit: version counter moves -> repack
            y2 = hip_linear.linear(x, lin.weight, lin.bias, params=True)
        assert len(calls) <= 2      # (1 when the stale pack was rebuilt by the batched repack of all registered packs)
        assert torch.allclose(y2, F.linear(x, lin.weight, lin.bias), rtol=1e-4, atol=1e-4)
    finally:
        hip_linear.pack_weight = orig


def test_without_the_private_engine_hooks_gradients_stay_on_the_main_stream():
    """A PyTorch build without torch._C._current_graph_task_id / the engine's queue_callback: no side stream, gradients
    published inside backward, same values."""
    from nsdp_amd import hip_linear
    lin, x = _layer(3, n=128, k=128)
    big = torch.randn(140000, 128, device=DEV, requires_grad=True)      # >= _OVERLAP_MIN_ROWS: would take the side stream
    hip_linear.linear(big, lin.weight, lin.bias, params=True).square().sum().backward()
    ref_w, ref_b = lin.weight.grad.clone(), lin.bias.grad.clone()
    lin.zero_grad()
    was = hip_linear._HAVE_ENGINE_HOOKS
    hip_linear._HAVE_ENGINE_HOOKS = False
    try:
        called = []
        orig = hip_linear._wgrad_deferred
        hip_linear._wgrad_deferred = lambda *a, **k: (called.append(1), orig(*a, **k))[1]
        hip_linear.linear(big, lin.weight, lin.bias, params=True).square().sum().backward()
        hip_linear._wgrad_deferred = orig
    finally:
        hip_linear._HAVE_ENGINE_HOOKS = was
    assert not called
    assert torch.allclose(lin.weight.grad, ref_w, rtol=1e-5, atol=1e-3) and torch.allclose(lin.bias.grad, ref_b, rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize("side", ["0", "1"])
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("k,n,M", [(128, 120, 4096), (200, 200, 40000), (3, 120, 4096)])
def test_shared_parameters_and_existing_grads_accumulate_in_the_kernel(side, dtype, k, n, M):
    """A parameter used several times in one graph (FlowArbitrary runs one encoder three times) and an existing `.grad`
    (flat-bucket views, micro-batch accumulation): the weight-gradient kernels add into the buffer that is already there
    (their `accumulate` flag), on the side stream and on the main stream, instead of through extra add kernels -- same sums
    as autograd's, and the `.grad` tensor object (a bucket view) is kept."""
    from nsdp_amd import hip_linear, precision
    torch.manual_seed(5)
    lin = torch.nn.Linear(k, n).to(DEV)
    xs = [torch.randn(M, k, device=DEV) for _ in range(3)]
    was = hip_linear._OVERLAP_WGRAD
    hip_linear._OVERLAP_WGRAD = side == "1"
    try:
        ref = torch.autograd.grad(sum(F.linear(x, lin.weight, lin.bias).square().sum() for x in xs), [lin.weight, lin.bias])
        tol = dict(rtol=2e-2, atol=2e-2 * float(ref[0].abs().max())) if dtype == "bf16" else dict(rtol=1e-4, atol=1e-4 * float(ref[0].abs().max()))
        with precision.storage(dtype):
            cast = (lambda t: t.to(torch.bfloat16)) if (dtype == "bf16" and k != 3) else (lambda t: t)
            # (a) three uses in one graph, .grad empty before
            sum(hip_linear.linear(cast(x), lin.weight, lin.bias, params=True).float().square().sum() for x in xs).backward()
            torch.cuda.synchronize()
            assert torch.allclose(lin.weight.grad, ref[0], **tol) and torch.allclose(lin.bias.grad, ref[1], **tol)
            # (b) a second pass into the existing .grad (kept as the same tensor objects: flat-bucket views stay views)
            gw, gb = lin.weight.grad, lin.bias.grad
            sum(hip_linear.linear(cast(x), lin.weight, lin.bias, params=True).float().square().sum() for x in xs).backward()
            torch.cuda.synchronize()
            assert lin.weight.grad is gw and lin.bias.grad is gb
            assert torch.allclose(gw, 2 * ref[0], **tol) and torch.allclose(gb, 2 * ref[1], **tol)
    finally:
        hip_linear._OVERLAP_WGRAD = was


def test_in_place_accumulation_launches_no_add_kernels():
    """The point of the accumulate flag: no ATen add per extra use of a parameter."""
    from nsdp_amd import hip_linear
    from torch.profiler import profile, ProfilerActivity
    if not hip_linear._PARAM_GRADS_DIRECT:
        pytest.skip("NSDP_PARAM_GRADS=autograd: autograd's own AccumulateGrad adds the uses up")
    torch.manual_seed(6)
    lin = torch.nn.Linear(128, 128).to(DEV)
    xs = [torch.randn(4096, 128, device=DEV) for _ in range(3)]

    def run():
        sum(hip_linear.linear(x, lin.weight, lin.bias, params=True).square().sum() for x in xs).backward()
    run()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU]) as prof:
        run()
        torch.cuda.synchronize()
    names = [e.name for e in prof.events()]
    assert names.count("aten::add_") == 0, names.count("aten::add_")      # (was: one per extra use and parameter)


@pytest.mark.parametrize("k,n,M,bias", [(3, 200, 70000, True), (4, 120, 40001, True), (3, 256, 8192, False)])
def test_k4_weight_gradient_with_recomputed_relu_mask(k, n, M, bias):
    """First layer of a position-encoding MLP (K = 3 / 4, fused output ReLU): the weight-gradient kernel recomputes the ReLU
    mask from the coordinates (nsdp_linear_wgrad_k4_remask_f32) instead of reading the [M, N] output back -- bit-identical to
    the masked-read path (same mask decisions, same summation), and right against autograd."""
    from nsdp_amd import hip_linear
    torch.manual_seed(M)
    lin = torch.nn.Linear(k, n, bias=bias).to(DEV)
    x = torch.randn(M, k, device=DEV)
    t = torch.randn(M, n, device=DEV)
    ref = torch.autograd.grad((F.relu(F.linear(x, lin.weight, lin.bias)) * t).sum(), list(lin.parameters()))
    got = []
    for on in (True, False):
        hip_linear.REMASK_K4 = on
        try:
            lin.zero_grad()
            (hip_linear.linear(x, lin.weight, lin.bias, relu_out=True, params=True) * t).sum().backward()
            torch.cuda.synchronize()
            got.append([p.grad.clone() for p in lin.parameters()])
        finally:
            hip_linear.REMASK_K4 = True
    for a, b in zip(*got):
        assert torch.equal(a, b)
    for a, r in zip(got[0], ref):
        assert torch.allclose(a, r, rtol=1e-4, atol=1e-4 * float(r.abs().max()))


@pytest.mark.parametrize("k,d,M,bias", [(3, 200, 200_003, True), (4, 256, 100_000, True), (3, 256, 65_536 + 17, False),
                                        (3, 200, 65_536, True)])
def test_k4_tail_weight_gradient_out_of_the_dx_gemm(k, d, M, bias):
    """Position-encoding MLP Linear(k, d) -> ReLU -> Linear(d, d) on coordinates that need no gradient: the second layer's dX
    GEMM forms the first layer's dW / db in its epilogue (nsdp_linear_bf16x3_k4tail_f32, ops.k4_tail) -- the [M, d] gradient of
    the hidden tensor is never written.  Against fp64 autograd and against the two-launch path (same mask decisions: the ReLU
    mask recomputed from the coordinates is the forward kernel's expression); deterministic; accumulates into existing grads."""
    from nsdp_amd import hip_linear
    from nsdp_amd.model import ops
    from test_model_gpu import _variant_trace
    if ops.PAIR_MASK or not hip_linear._PARAM_GRADS_DIRECT or not hip_linear.K4_LINK:
        pytest.skip("knob run: the link between the two layers is off")
    torch.manual_seed(M + d)
    seq = torch.nn.Sequential(torch.nn.Linear(k, d, bias=bias), torch.nn.ReLU(), torch.nn.Linear(d, d)).to(DEV)
    x = torch.randn(M, k, device=DEV)
    t = torch.randn(M, d, device=DEV)
    seq64 = torch.nn.Sequential(torch.nn.Linear(k, d, bias=bias), torch.nn.ReLU(), torch.nn.Linear(d, d)).to(DEV).double()
    seq64.load_state_dict(seq.state_dict())
    (seq64(x.double()) * t.double()).sum().backward()
    ref = [p.grad for p in seq64.parameters()]
    got = {}
    trace = {}
    for mode in ("tail", "two", "tail", "nolink"):        # fused epilogue / linked, two launches / again / plain autograd path
        hip_linear.K4_TAIL, hip_linear.K4_LINK = mode == "tail", mode != "nolink"
        try:
            seq.zero_grad()
            with _variant_trace() as names:
                (ops.mlp2(x, seq) * t).sum().backward()
            torch.cuda.synchronize()
            got.setdefault(mode, []).append([p.grad.clone() for p in seq.parameters()])
            trace[mode] = names
        finally:
            hip_linear.K4_TAIL = hip_linear.K4_LINK = True
    fused = hip_linear._USE_X3          # (NSDP_BF16X3=0: no kernel with the tail form; the linked two-launch path runs instead)
    assert any("k4tail" in n for n in trace["tail"]) == fused, trace["tail"]
    assert not any("k4tail" in n for n in trace["two"]) and not any("k4tail" in n for n in trace["nolink"])
    for a, b in zip(*got["tail"]):                     # the same launch twice: bit-equal
        assert torch.equal(a, b)
    for a, b in zip(got["two"][0], got["nolink"][0]):  # the linked two-launch form IS the autograd path's kernels
        assert torch.equal(a, b)
    for a, b, r in zip(got["tail"][0], got["two"][0], ref):
        s = float(r.abs().max())
        assert float((a.double() - r).abs().max()) <= 2e-5 * s, (a.shape, float((a.double() - r).abs().max()), s)
        assert float((b.double() - r).abs().max()) <= 2e-5 * s
    # existing gradients: the launch adds to them
    before = [p.grad.clone() for p in seq.parameters()]
    (ops.mlp2(x, seq) * t).sum().backward()
    for p, b0, r in zip(seq.parameters(), before, ref):
        assert float((p.grad.double() - 2 * r).abs().max()) <= 4e-5 * float(r.abs().max())


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_batched_weight_gradient_reduce_is_bit_equal_to_the_one_call_form(dtype):
    """nsdp_linear_wgrad_bf16x3_partials_f32 / nsdp_linear_wgrad_bf16_partials + one *_reduce_batched launch over several layers
    against the one-call forms: the SAME bits, fresh targets and accumulated ones, with and without a bias gradient -- and through
    the autograd wrapper: a backward pass with NSDP_WGRAD_BATCH_REDUCE = 16 / 3 / 0 leaves identical gradients, including a
    parameter used twice in the graph (its pending reduction must land before the second use accumulates into it)."""
    import ctypes
    from nsdp_amd import _lib, hip_linear, precision
    L = _lib.lib()
    torch.manual_seed(11)
    shapes = [(40000, 200, 200, True), (8192, 128, 120, False), (70001, 256, 256, True), (4096, 120, 128, True)]
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float32
    p = (lambda t: ctypes.c_void_p(t.data_ptr()))
    descs, keep, want = [], [], []
    Desc = hip_linear._ReduceDescB16 if dtype == "bf16" else hip_linear._ReduceDesc
    for i, (M, N, K, bias) in enumerate(shapes):
        dy = torch.randn(M, N, device=DEV).to(tdt)
        x = torch.randn(M, K, device=DEV).to(tdt)
        acc = i % 2                                        # every other layer accumulates into what is there
        base_w, base_b = torch.randn(N, K, device=DEV), torch.randn(N, device=DEV)
        outs = []
        for form in ("one", "batched"):
            dw, db = base_w.clone(), (base_b.clone() if bias else None)
            if dtype == "bf16":
                L.nsdp_linear_wgrad_bf16_workspace_bytes.restype = ctypes.c_size_t
                nb = int(L.nsdp_linear_wgrad_bf16_workspace_bytes(ctypes.c_longlong(M), N, K))
            else:
                L.nsdp_linear_wgrad_bf16x3_workspace_bytes.restype = ctypes.c_size_t
                nb = int(L.nsdp_linear_wgrad_bf16x3_workspace_bytes(ctypes.c_longlong(M), N, K))
            ws = torch.empty(nb // 4, device=DEV)
            args = (p(dy), p(x), None, 0, p(dw), p(db) if bias else None, ctypes.c_longlong(M), N, K, acc, p(ws), ctypes.c_size_t(nb))
            if form == "one":
                fn = L.nsdp_linear_wgrad_bf16 if dtype == "bf16" else L.nsdp_linear_wgrad_bf16x3_f32
                _lib.check(fn(*args, _lib.stream_ptr()), "one-call weight gradient")
            else:
                d = Desc()
                fn = L.nsdp_linear_wgrad_bf16_partials if dtype == "bf16" else L.nsdp_linear_wgrad_bf16x3_partials_f32
                _lib.check(fn(*args, ctypes.byref(d), _lib.stream_ptr()), "partials")
                descs.append(d)
                keep.append(ws)
            outs.append((dw, db))
        want.append(outs)
    arr = (Desc * len(descs))(*descs)
    fn = L.nsdp_wgrad_bf16_reduce_batched if dtype == "bf16" else L.nsdp_wgrad_bf16x3_reduce_batched
    _lib.check(fn(arr, len(descs), _lib.stream_ptr()), "batched reduce")
    torch.cuda.synchronize()
    for (one, bat), shp in zip(want, shapes):
        assert torch.equal(one[0], bat[0]), shp
        if shp[3]:
            assert torch.equal(one[1], bat[1]), shp

    # through autograd: three layers on the side stream, the first one used TWICE (large rows, then a few rows)
    lins = [torch.nn.Linear(200, 200).to(DEV), torch.nn.Linear(200, 128).to(DEV), torch.nn.Linear(128, 128).to(DEV)]
    x_big, x_small = torch.randn(40000, 200, device=DEV), torch.randn(16, 200, device=DEV)
    grads = {}
    was_o, was_b = hip_linear._OVERLAP_WGRAD, hip_linear.BATCH_REDUCE
    hip_linear._OVERLAP_WGRAD = True
    try:
        for batch in (16, 3, 0):
            hip_linear.BATCH_REDUCE = batch
            for l in lins:
                l.weight.grad = l.bias.grad = None
            with precision.storage(dtype):
                c = (lambda t: t.to(tdt))
                h = hip_linear.linear(c(x_big), lins[0].weight, lins[0].bias, relu_out=True, params=True)
                h = hip_linear.linear(h, lins[1].weight, lins[1].bias, relu_out=True, params=True)
                h = hip_linear.linear(h, lins[2].weight, lins[2].bias, params=True)
                g = hip_linear.linear(c(x_small), lins[0].weight, lins[0].bias, params=True)
                (h.float().square().sum() + g.float().square().sum()).backward()
            torch.cuda.synchronize()
            grads[batch] = [t.grad.clone() for l in lins for t in (l.weight, l.bias)]
    finally:
        hip_linear._OVERLAP_WGRAD, hip_linear.BATCH_REDUCE = was_o, was_b
    for a, b, c3 in zip(grads[16], grads[3], grads[0]):
        assert torch.equal(a, c3) and torch.equal(b, c3)


def _pos_mlp_gather(M, d, form, seed):
    """A gather tuple of hip_linear's init_gather protocol for M rows: form 1 = (queries per point, key table), 2 = one difference table."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    k, nsrc = 16, 100
    shapes = max(M // 4096, 1)
    rps = (M + shapes - 1) // shapes
    rps += (-rps) % k
    gk = torch.randn(shapes * nsrc, d, generator=g).to(DEV)
    gidx = torch.randint(0, nsrc, (M,), generator=g, dtype=torch.int32).to(DEV)
    if form == 2:
        return (None, 1, gk, gidx, rps, nsrc)
    gq = torch.randn((M + k - 1) // k, d, generator=g).to(DEV)
    return (gq, k, gk, gidx, rps, nsrc)


@pytest.mark.parametrize("k,d,M,bias,form", [(3, 200, 200_003, True, 2), (4, 256, 100_000, True, 1), (3, 256, 65_536 + 17, False, 0),
                                             (3, 128, 70_000, True, 1), (3, 200, 40_000, True, 0), (4, 128, 262_144, True, 2),
                                             (3, 256, 51_200, True, 1)])
def test_position_encoding_mlp_without_its_hidden_tensor(k, d, M, bias, form, monkeypatch):
    """hip_linear.pos_mlp: Linear(k, d) -> ReLU -> Linear(d, d) on coordinates that need no gradient, the hidden tensor recomputed
    inside the second layer's GEMM (nsdp_linear_bf16x3_h0_f32) and inside its weight gradient (nsdp_linear_wgrad_bf16x3_h0_f32).
    The producers evaluate the K = 4 layer with the forward kernel's own expression: output and all four parameter gradients are
    BIT-equal to the two-layer path's, with and without the gathered addend, on whole and ragged row counts."""
    from nsdp_amd import hip_linear
    from nsdp_amd.model import ops
    from test_model_gpu import _variant_trace
    if (ops.PAIR_MASK or not hip_linear._PARAM_GRADS_DIRECT or not hip_linear.K4_LINK or not hip_linear.REMASK_K4
            or not hip_linear._USE_X3 or not hip_linear.H0_RECOMPUTE):
        pytest.skip("knob run: the recomputed form is off")
    monkeypatch.setattr(hip_linear, "H0_RECOMPUTE", 2)      # (every shape the kernels support, not only where it pays)
    torch.manual_seed(M + d + form)
    seq = torch.nn.Sequential(torch.nn.Linear(k, d, bias=bias), torch.nn.ReLU(), torch.nn.Linear(d, d)).to(DEV)
    x = torch.randn(M, k, device=DEV)
    if k == 3:
        x = torch.nn.functional.pad(x, (0, 1))      # (ops.relative_coords hands the rows over zero-padded)
    t = torch.randn(M, d, device=DEV)
    gather = _pos_mlp_gather(M, d, form, M) if form else None

    def two_layers():
        tl = ops.k4_tail(x, seq)
        h = ops.linear(x, seq[0], relu=True, tail_src=tl)
        return ops.linear(h, seq[2], init_gather=gather, tail_dst=tl)

    with torch.no_grad(), _variant_trace() as names:
        y_new = ops.pos_mlp(x, seq, init_gather=gather)
    assert y_new is not None and any(n.startswith("linear_bf16x3<") and n.split("<")[1].split(",")[2] == "3" for n in names), names
    with torch.no_grad():
        y_old = two_layers()
    assert torch.equal(y_new, y_old)
    grads = []
    for fn in (lambda: ops.pos_mlp(x, seq, init_gather=gather), two_layers, lambda: ops.pos_mlp(x, seq, init_gather=gather)):
        seq.zero_grad()
        y = fn()
        assert torch.equal(y, y_old)
        (y * t).sum().backward()
        torch.cuda.synchronize()
        grads.append([p.grad.clone() for p in seq.parameters()])
    for a, b, c in zip(*grads):
        assert torch.equal(a, b) and torch.equal(a, c), (a.shape, float((a - b).abs().max()))
    # existing gradients: the launches add to them
    (ops.pos_mlp(x, seq, init_gather=gather) * t).sum().backward()
    for p, g in zip(seq.parameters(), grads[0]):
        assert torch.allclose(p.grad, 2 * g, rtol=1e-5, atol=1e-5 * float(g.abs().max()))


def test_position_encoding_mlp_falls_back_where_the_recomputed_form_does_not_apply(monkeypatch):
    from nsdp_amd import hip_linear
    from nsdp_amd.model import ops
    if not hip_linear.H0_RECOMPUTE:
        pytest.skip("knob run: the recomputed form is off")
    monkeypatch.setattr(hip_linear, "H0_RECOMPUTE", 2)
    seq = torch.nn.Sequential(torch.nn.Linear(3, 200), torch.nn.ReLU(), torch.nn.Linear(200, 200)).to(DEV)
    x = torch.randn(70_000, 3, device=DEV)
    assert ops.pos_mlp(x.clone().requires_grad_(True), seq) is None            # coordinates that need a gradient
    assert ops.pos_mlp(x[:1000], seq) is None                                  # too few rows for the bf16x3 kernels
    small = torch.nn.Sequential(torch.nn.Linear(3, 32), torch.nn.ReLU(), torch.nn.Linear(32, 32)).to(DEV)
    assert ops.pos_mlp(x, small) is None                                       # one k block
    y = ops.mlp2(x.clone().requires_grad_(True), seq)                          # ... and mlp2 still answers
    assert y.shape == (70_000, 200) and y.requires_grad


@pytest.mark.parametrize("M,N,K,mask,relu_x", [(800, 256, 256, False, False), (777, 200, 120, True, False), (33, 40, 24, False, True),
                                              (2047, 120, 256, True, True), (1, 16, 16, False, False), (1999, 72, 200, False, False)])
def test_output_stationary_weight_gradient_for_few_rows(M, N, K, mask, relu_x):
    """linear_wgrad_direct_kernel (csrc/gemm.hip): up to 2048 rows a workgroup owns a 32 x 32 block of dW and walks all rows -- one
    launch, no partial sums.  Against fp64 on whole and ragged shapes, with the ReLU mask / input ReLU, bias gradient, in-place
    accumulation; bit-equal between two runs (fixed summation order)."""
    from nsdp_amd import hip_linear
    from test_model_gpu import _variant_trace
    g = torch.Generator(device="cpu").manual_seed(M * 7 + N + K)
    dy = torch.randn(M, N, generator=g).to(DEV)
    x = torch.randn(M, K, generator=g).to(DEV)
    mk = torch.randn(M, N, generator=g).to(DEV) if mask else None
    with _variant_trace() as names:
        dw, db = hip_linear._wgrad(dy, x, mk, relu_x, True)
    if os.environ.get("NSDP_WGRAD_DIRECT_ROWS", "2048") != "0":
        assert any(n.startswith("linear_wgrad_direct") for n in names), names
    dyr = dy.double() * (mk > 0) if mask else dy.double()
    xr = x.double().clamp_min(0) if relu_x else x.double()
    rw, rb = dyr.t() @ xr, dyr.sum(0)
    assert float((dw.double() - rw).abs().max()) <= 2e-6 * (float(rw.abs().max()) + 1e-6) * max(1.0, M ** 0.5 / 8)
    assert float((db.double() - rb).abs().max()) <= 2e-6 * (float(rb.abs().max()) + 1e-6) * max(1.0, M ** 0.5 / 8)
    dw2, db2 = hip_linear._wgrad(dy, x, mk, relu_x, True)
    assert torch.equal(dw, dw2) and torch.equal(db, db2)
    acc_w, acc_b = dw.clone(), db.clone()
    got = hip_linear._wgrad(dy, x, mk, relu_x, True, out=(acc_w, acc_b))
    assert got[0] is acc_w
    assert torch.allclose(acc_w, 2 * dw, rtol=1e-6, atol=1e-6 * float(dw.abs().max()))
    assert torch.allclose(acc_b, 2 * db, rtol=1e-6, atol=1e-6 * float(db.abs().max()))
