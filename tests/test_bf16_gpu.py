"""bf16-STORAGE path (BASELINE config 3): the bf16 kernels against fp64 references evaluated on the SAME bf16-rounded
inputs (so the only difference is accumulation order and the final rounding to bf16), and the model-level deviation
from the fp32 reference fixtures (no reference bar exists for bf16 -- the numbers are reported and loosely bounded)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import batchnorm_three_launch, build_product, fixture_setup, l2_err, model_cfg, restore_model, run_forward, snapshot_model, to_dev
from nsdp_amd import synth

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
BF = torch.bfloat16


def _rnd(shape, g, scale=1.0):
    return (torch.randn(*shape, generator=g) * scale).to(BF)


LIN = [  # (M, K, N, bias, relu_in, relu_out, residual, mask, out_mask, out_f32)
    (1, 8, 4, True, False, False, False, False, False, False),
    (37, 120, 120, True, False, True, False, False, False, False),
    (1000, 128, 256, False, False, False, True, False, False, False),
    (5131, 200, 200, True, False, True, False, False, False, False),
    (5131, 200, 200, False, False, False, False, True, False, False),
    (5131, 200, 200, False, False, False, False, False, True, False),
    (777, 200, 128, True, True, False, True, False, False, False),
    (3000, 128, 200, False, False, False, True, True, True, False),
    (4099, 256, 256, True, False, False, False, True, False, False),
    (4099, 256, 120, True, False, True, False, False, False, False),
    (300, 120, 256, True, True, True, False, False, False, False),
    (2000, 128, 3, True, True, False, False, False, False, True),
    (129, 24, 12, True, False, False, True, False, True, False),
    (70000, 128, 128, True, True, True, True, False, False, False),
]
# every kernel variant (n tiles 4 / 8 / 13 / 16 x k blocks <= 4 / <= 8 x mask) with several tiles per wave: a prefetch
# that delivers the wrong tile's rows shows up here (70 000 rows > 256 workgroups x 128 rows)
for _n in (64, 128, 200, 256):
    for _k in (128, 200):
        for _m in (False, True):
            LIN.append((70000, _k, _n, True, False, False, False, _m, False, False))


@pytest.mark.parametrize("M,K,N,bias,relu_in,relu_out,res,mask,omask,f32", LIN)
def test_linear_bf16_forward_kernel(M, K, N, bias, relu_in, relu_out, res, mask, omask, f32):
    from nsdp_amd import hip_linear_bf16 as hb
    g = torch.Generator().manual_seed(M * 131 + K * 7 + N)
    x = _rnd((M, K), g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g) if bias else None
    r = _rnd((M, N), g) if res else None
    mk = _rnd((M, K), g).clamp_min(0) if mask else None
    om = _rnd((M, N), g).clamp_min(0) if omask else None
    wp, _ = hb.pack_weight_b16(w.to(DEV), True, False)
    dev = lambda t: None if t is None else t.to(DEV)
    y = hb.run(dev(x), wp, N, dev(b), dev(r), dev(mk), dev(om), relu_in, relu_out, out_f32=f32)
    assert y.dtype is (torch.float32 if f32 else BF)
    # reference: the same bf16 inputs and bf16-rounded weights, fp64 arithmetic
    xi = x.double()
    if mk is not None:
        xi = xi * (mk > 0)
    if relu_in:
        xi = F.relu(xi)
    ref = xi @ w.to(BF).double().t()
    if b is not None:
        ref = ref + b.double()
    if r is not None:
        ref = ref + r.double()
    if relu_out:
        ref = F.relu(ref)
    if om is not None:
        ref = ref * (om > 0)
    err = (y.double().cpu() - ref).abs()
    tol = 1e-5 * (1 + ref.abs()) if f32 else 2 ** -8 * ref.abs() + 1e-5      # one bf16 rounding of the result
    assert bool((err <= tol).all()), float((err - tol).max())


WG = [(4096, 120, 120, False, False, True), (5000, 200, 200, True, False, True), (4099, 256, 256, False, True, True),
      (33, 128, 200, True, True, False), (70001, 200, 128, False, False, True), (262144, 128, 128, True, True, True),
      (3200, 256, 120, False, False, True), (31, 8, 8, False, False, True),
      # (the transpose-read kernel: ragged last slab, a mask at 256 x 256 -- three images per ring slot --, one slab more
      # than a workgroup's share, widths that are multiples of 8 but not of 16; and the shapes that stay on the
      # transposing kernel: widths that are not multiples of 8, fewer than 64 rows)
      (4097, 256, 256, True, True, True), (65, 200, 200, True, False, True), (40000 + 17, 72, 248, False, True, True),
      (9000, 250, 122, True, False, True), (63, 64, 64, False, False, True)]


@pytest.mark.parametrize("old_kernel", [False, True])
@pytest.mark.parametrize("M,N,K,mask,relu_x,want_db", WG)
def test_linear_bf16_wgrad_kernel(M, N, K, mask, relu_x, want_db, old_kernel):
    """Both weight-gradient kernels of the bf16 path (operands by hardware transpose reads from the row-major slab images;
    explicit transposition into fragment images: nsdp_debug_set(7, 8)) against fp64 on the same bf16 inputs."""
    from nsdp_amd import _lib, hip_linear_bf16 as hb
    g = torch.Generator().manual_seed(M + 3 * N + 5 * K)
    dy, x = _rnd((M, N), g), _rnd((M, K), g)
    mk = _rnd((M, N), g).clamp_min(0) if mask else None
    _lib.lib().nsdp_debug_set(7, 8 if old_kernel else 0)
    try:
        dw, db = hb.wgrad(dy.to(DEV), x.to(DEV), None if mk is None else mk.to(DEV), relu_x, want_db)
        torch.cuda.synchronize()
    finally:
        _lib.lib().nsdp_debug_set(7, 0)
    dyd = dy.double() * (mk > 0) if mask else dy.double()
    xd = F.relu(x.double()) if relu_x else x.double()
    ref = dyd.t() @ xd
    scale = float(ref.abs().max()) + 1e-6
    # exact bf16 products, fp32 accumulation over M rows
    assert float((dw.double().cpu() - ref).abs().max()) <= 3e-6 * scale * max(1.0, (M / 4096) ** 0.5) + 1e-6
    if want_db:
        rb = dyd.sum(0)
        assert float((db.double().cpu() - rb).abs().max()) <= 3e-6 * (float(rb.abs().max()) + 1) * max(1.0, (M / 4096) ** 0.5)
    else:
        assert db is None


def test_linear_bf16_autograd_matches_reference():
    """The autograd wrapper in bf16 storage mode: forward, dX (with ReLU masks) and direct parameter gradients."""
    from nsdp_amd import hip_linear, precision
    torch.manual_seed(3)
    lin1, lin2 = torch.nn.Linear(200, 128).to(DEV), torch.nn.Linear(128, 200).to(DEV)
    x = torch.randn(9000, 200, device=DEV).to(BF).requires_grad_(True)
    with precision.storage(BF):
        h = hip_linear.linear(x, lin1.weight, lin1.bias, relu_in=True, relu_out=True, params=True)
        y = hip_linear.linear(h, lin2.weight, lin2.bias, residual=x, params=True)
        assert h.dtype is BF and y.dtype is BF
        y.float().square().sum().backward()
    got = (x.grad.float(), lin1.weight.grad, lin1.bias.grad, lin2.weight.grad, lin2.bias.grad)
    xr = x.detach().float().requires_grad_(True)
    ws = [p.detach().clone().requires_grad_(True) for p in (lin1.weight, lin1.bias, lin2.weight, lin2.bias)]
    hr = F.relu(F.linear(F.relu(xr), ws[0].to(BF).float(), ws[1])).to(BF).float()
    yr = (F.linear(hr, ws[2].to(BF).float(), ws[3]) + xr).to(BF).float()
    yr.square().sum().backward()
    ref = (xr.grad,) + tuple(w.grad for w in ws)
    for a, b, name in zip(got, ref, ("dx", "dW1", "db1", "dW2", "db2")):
        rel = float((a - b).norm() / (b.norm() + 1e-12))
        assert rel <= 1.5e-2, (name, rel)         # bf16 rounding of h, dh, dy between the layers


@pytest.mark.parametrize("mtype", ["forward", "backward", "arbitrary"])
def test_bf16_storage_model_vs_fp32_reference(mtype):
    """Eval forward and one train step of the whole model with bf16 storage against the fp32 reference fixtures."""
    from nsdp_amd import precision
    from nsdp_amd.model import optimizer_factory
    fx, cfg, seed, data = fixture_setup("tiny_" + mtype, mtype)
    model, train_fn, _ = build_product(cfg, seed, DEV)
    with precision.storage(BF):
        model.eval()
        with torch.no_grad():
            out = run_forward(model, cfg, to_dev(data, DEV))
        assert out.dtype is torch.float32
        l2 = l2_err(out.cpu().numpy(), fx["eval_out"])
        model.train()
        _, opt = optimizer_factory({"optimizer": "Adam", "lr": 5e-4}, model.parameters())
        loss = train_fn(model, opt, to_dev(data, DEV), cfg)
    print(f"\\nbf16 storage, tiny_{mtype}: eval L2 vs fp32 reference {l2:.3e}; train loss {loss:.6f} vs {float(fx['train_loss']):.6f}")
    # 'arbitrary' feeds the first network's bf16-perturbed OUTPUT POINTS to the second network's farthest-point sampling
    # and kNN: index selection is discontinuous, so with these untrained procedural weights a 1e-2 perturbation of the
    # canonical points selects different anchors and moves individual outputs by O(1) -- bounded loosely, reported above
    assert l2 <= (0.3 if mtype == "arbitrary" else 3e-2), l2
    assert abs(loss - float(fx["train_loss"])) <= (0.12 if mtype == "arbitrary" else 0.05) * abs(float(fx["train_loss"])) + 1e-3
    rels = []
    for k, p in model.named_parameters():
        if p.grad is None:
            continue
        assert p.grad.dtype is torch.float32 and bool(torch.isfinite(p.grad).all()), k
        gn = float(fx["grad_norm/" + k])
        if gn > 1e-4:
            rels.append(abs(float(p.grad.double().norm()) - gn) / gn)
    print(f"gradient-norm deviation from the fp32 reference: median {np.median(rels):.3e}, max {np.max(rels):.3e}")
    assert np.median(rels) <= (0.25 if mtype == "arbitrary" else 0.05)


def _close_bf16(a, b, name, atol):
    """Two bf16 tensors computed with identical fp32 arithmetic up to summation order: at most an ulp apart."""
    a, b = a.float(), b.float()
    err = (a - b).abs()
    tol = 2 ** -7 * b.abs() + atol
    assert bool((err <= tol).all()), (name, float((err - tol).max()))


@pytest.mark.parametrize("B,n,N,k,d,qb,glob", [(2, 37, 50, 10, 120, 0, False), (3, 1000, 100, 7, 200, 1, True),
                                                (2, 64, 64, 16, 256, 0, False), (1, 5000, 100, 7, 200, 1, True)])
def test_attention_glue_bf16_native_vs_cast_reference(B, n, N, k, d, qb, glob):
    """The bf16-storage instantiation of the attention kernels against the fp32 kernels run on casts of the same bf16
    inputs (hip_attention.NATIVE_BF16 = False): forward values and every gradient."""
    from nsdp_amd import hip_attention as ha
    g = torch.Generator().manual_seed(B * 1000 + n)
    mk = lambda *s: (torch.randn(*s, generator=g)).to(BF).to(DEV)
    idx = torch.randint(0, N, (B, n, k), generator=g, dtype=torch.int32).to(DEV)
    base = dict(q=mk(B, 1 if qb else n, d), kf=mk(B, N, d), vf=mk(B, N, d), pos=mk(B, n, k, d), res=mk(B, n, d),
                a_g=mk(B, d) if glob else None, v_g=mk(B, d) if glob else None, w=mk(B, n, d))
    outs = []
    for native in (True, False):
        ha.NATIVE_BF16 = native
        try:
            t = {kk: (None if v is None else v.clone().requires_grad_(True)) for kk, v in base.items() if kk != "w"}
            link = ha.pos_grad_link()
            u = ha.attn_pre(t["q"], t["kf"], t["pos"], idx, link)
            a = u * 0.5                                      # stands in for the gamma MLP
            y = ha.attn_post(a, t["vf"], t["pos"], idx, a_g=t["a_g"], v_g=t["v_g"], residual=None if glob else t["res"],
                             link=link)
            (y.float() * base["w"].float()).sum().backward()
            outs.append((u.detach(), y.detach(), {kk: v.grad for kk, v in t.items() if v is not None and v.grad is not None}))
        finally:
            ha.NATIVE_BF16 = True
    (u1, y1, g1), (u0, y0, g0) = outs
    assert u1.dtype is BF and y1.dtype is BF
    _close_bf16(u1, u0, "u", 1e-6)
    _close_bf16(y1, y0, "y", 1e-3)
    assert set(g1) == set(g0)
    # dq = sum_j du_j is analytically ~0 when q is per point (the softmax gradient sums to zero over the neighbours): it
    # consists of bf16 rounding noise of the du_j, so every gradient is judged against the scale of the du-sized ones
    floor = float(g0["kf"].float().abs().max())
    for kk in g1:
        scale = max(float(g0[kk].float().abs().max()), floor) + 1e-6
        rel = float((g1[kk].float() - g0[kk].float()).abs().max()) / scale
        assert rel <= 2e-2, (kk, rel)


@pytest.mark.parametrize("R,C,addend,relu", [(3200, 256, False, True), (65536, 120, True, False), (1000, 256, True, True)])
def test_batchnorm_bf16_native_vs_cast_reference(R, C, addend, relu):
    from nsdp_amd import hip_batchnorm as hbn
    from nsdp_amd.model import ops
    g = torch.Generator().manual_seed(R + C)
    x0 = (torch.randn(R, C, generator=g) * 2 + 3).to(BF).to(DEV)
    a0 = torch.randn(R, C, generator=g).to(BF).to(DEV) if addend else None
    w = torch.randn(R, C, generator=g).to(DEV)
    res = []
    for native in (True, False):
        hbn.NATIVE_BF16 = native
        try:
            torch.manual_seed(0)
            bn = torch.nn.BatchNorm1d(C).to(DEV)
            with torch.no_grad():
                bn.weight.uniform_(0.5, 1.5)
                bn.bias.uniform_(-0.5, 0.5)
            x = x0.clone().requires_grad_(True)
            a = None if a0 is None else a0.clone().requires_grad_(True)
            y = ops.batch_norm(x, bn, addend=a, relu=relu)
            (y.float() * w).sum().backward()
            res.append((y.detach(), x.grad, bn.weight.grad, bn.bias.grad, bn.running_mean.clone(), bn.running_var.clone()))
        finally:
            hbn.NATIVE_BF16 = True
    n1, n0 = res
    assert n1[0].dtype is BF and n1[1].dtype is BF
    _close_bf16(n1[0], n0[0], "y", 1e-3)
    for i, name in ((1, "dx"), (2, "dgamma"), (3, "dbeta"), (4, "running_mean"), (5, "running_var")):
        scale = float(n0[i].float().abs().max()) + 1e-6
        assert float((n1[i].float() - n0[i].float()).abs().max()) <= 1.5e-2 * scale, name


def test_batchnorm_large_mean_small_std_is_stable():
    """Shifted sums: a channel with |mean| >> std must not lose its variance to cancellation (fp32 path)."""
    from nsdp_amd.model import ops
    torch.manual_seed(1)
    x = (torch.randn(8192, 8, device=DEV) * 1e-3 + 100.0).requires_grad_(True)
    bn = torch.nn.BatchNorm1d(8).to(DEV)
    ref = torch.nn.BatchNorm1d(8).to(DEV)
    y = ops.batch_norm(x, bn)
    yr = ref(x.detach().double().float())
    ref64 = (x.detach().double() - x.detach().double().mean(0)) / (x.detach().double().var(0, unbiased=False) + 1e-5).sqrt()
    assert float((y.double() - ref64).abs().max()) < 2e-2
    assert float((bn.running_var - (0.9 + 0.1 * x.detach().double().var(0)).float()).abs().max()) < 1e-7


def test_batchnorm_momentum_none_is_cumulative_average():
    from nsdp_amd.model import ops
    torch.manual_seed(2)
    bn = torch.nn.BatchNorm1d(16, momentum=None).to(DEV)
    ref = torch.nn.BatchNorm1d(16, momentum=None).to(DEV)
    for i in range(3):
        x = torch.randn(512, 16, device=DEV) + i
        ops.batch_norm(x, bn)
        ref(x)
    assert torch.allclose(bn.running_mean, ref.running_mean, atol=1e-5)
    assert torch.allclose(bn.running_var, ref.running_var, atol=1e-5)
    assert int(bn.num_batches_tracked) == 3


@pytest.mark.parametrize("M,N,K3,relu,mask", [(5000, 200, True, True, True), (70001, 120, True, False, False),
                                               (4096, 256, False, False, False)])
def test_k4_bf16_kernels(M, N, K3, relu, mask):
    from nsdp_amd import hip_linear_bf16 as hb
    g = torch.Generator().manual_seed(M + N)
    x = torch.randn(M, 4, generator=g)
    if K3:
        x[:, 3] = 0
    w = torch.randn(N, 4, generator=g)
    b = torch.randn(N, generator=g)
    y = hb.k4_forward(x.to(DEV), w.to(DEV), b.to(DEV), relu)
    ref = x.double() @ w.double().t() + b.double()
    ref = F.relu(ref) if relu else ref
    assert bool(((y.double().cpu() - ref).abs() <= 2 ** -8 * ref.abs() + 1e-5).all())
    dy = _rnd((M, N), g)
    mk = y if mask else None
    dw, db = hb.k4_wgrad(dy.to(DEV), x.to(DEV), mk, False, True)
    dyd = dy.double() * (y.double().cpu() > 0) if mask else dy.double()
    rw, rb = dyd.t() @ x.double(), dyd.sum(0)
    assert float((dw.double().cpu() - rw).abs().max()) <= 1e-5 * (float(rw.abs().max()) + 1) * max(1.0, (M / 4096) ** 0.5)
    assert float((db.double().cpu() - rb).abs().max()) <= 1e-5 * (float(rb.abs().max()) + 1) * max(1.0, (M / 4096) ** 0.5)


@pytest.mark.parametrize("mtype", ["forward", "backward"])
def test_bf16_native_kernels_agree_with_the_cast_reference_in_the_model(mtype):
    """The whole model with bf16 storage, once through the native bf16 kernels and once through the fp32 kernels on
    casts (same storage semantics: every tensor rounded to bf16 at the same places, fp32 arithmetic in between): eval
    output and one train step.  Differences are accumulation-order noise (1e-6 relative) that flips individual bf16
    roundings and grows layer by layer -- measured per op: 1e-5 after the first attention block, 1e-4 after the first set
    abstraction, 1e-3 in the 256-wide blocks, ~1e-2 at the output after 173 ops; no jump at any kernel."""
    from nsdp_amd import hip_attention, hip_batchnorm, hip_linear_bf16, precision
    from nsdp_amd.model import optimizer_factory
    fx, cfg, seed, data = fixture_setup("tiny_" + mtype, mtype)
    res = []
    for native in (True, False):
        hip_attention.NATIVE_BF16 = hip_batchnorm.NATIVE_BF16 = hip_linear_bf16.NATIVE = native
        try:
            model, train_fn, _ = build_product(cfg, seed, DEV)
            with torch.no_grad():      # weights exactly representable in bf16: both paths then multiply by the same numbers
                for prm in model.parameters():
                    prm.copy_(prm.to(BF).float())
            with precision.storage(BF):
                model.eval()
                with torch.no_grad():
                    out = run_forward(model, cfg, to_dev(data, DEV)).cpu().numpy()
                model.train()
                _, opt = optimizer_factory({"optimizer": "Adam", "lr": 5e-4}, model.parameters())
                loss = train_fn(model, opt, to_dev(data, DEV), cfg)
            res.append((out, loss, {k: p.grad.double().norm().item() for k, p in model.named_parameters() if p.grad is not None}))
        finally:
            hip_attention.NATIVE_BF16 = hip_batchnorm.NATIVE_BF16 = hip_linear_bf16.NATIVE = True
    (o1, l1, g1), (o0, l0, g0) = res
    l2 = l2_err(o1, o0)
    rel = [abs(g1[k] - g0[k]) / g0[k] for k in g0 if g0[k] > 1e-4]
    print(f"\\nnative vs cast reference, tiny_{mtype}: eval L2 {l2:.2e}, loss {l1:.6f} / {l0:.6f}, "
          f"gradient norms: median {np.median(rel):.2e} max {np.max(rel):.2e}")
    assert l2 <= 3e-2, l2
    assert abs(l1 - l0) <= 1e-2 * abs(l0)
    assert np.median(rel) <= 3e-2


@pytest.mark.parametrize("B,rows,N,d", [(3, 7001, 100, 200), (2, 57344, 100, 200), (1, 33, 8, 8), (4, 5000, 128, 256)])
def test_scatter_as_gemm_onehot(B, rows, N, d):
    """table[b][a] = sum of the rows with idx == a, computed on the matrix cores against the one-hot index matrix."""
    from nsdp_amd import hip_attention as ha
    g = torch.Generator().manual_seed(rows)
    src = torch.randn(B, rows, d, generator=g).to(BF).to(DEV)
    idx = torch.randint(0, N, (B, rows), generator=g, dtype=torch.int32).to(DEV)
    if N >= 100:
        idx[:, :50] = N - 1                      # a hot row and (below) an empty one
        idx[idx == 3] = 4
    table = ha.onehot_scatter(src, idx, N)
    ref = torch.zeros(B, N, d, dtype=torch.float64, device=DEV)
    ref.scatter_add_(1, idx.long().unsqueeze(-1).expand(B, rows, d), src.double())
    assert float((table.double() - ref).abs().max()) <= 2e-6 * (float(ref.abs().max()) + 1) * max(1.0, (rows / 4096) ** 0.5)
    if N >= 100:
        assert float(table[:, 3].abs().max()) == 0.0
    assert torch.equal(table, ha.onehot_scatter(src, idx, N))          # deterministic (no atomics)


def _step_grads(cfg, data, mode, scale_inputs=1.0):
    from nsdp_amd import precision
    from nsdp_amd.model.utils import compute_l2_error
    model, _, _ = build_product(cfg, 3232, DEV)
    model.train()
    d = dict(data)
    if scale_inputs != 1.0:
        d["surface_samples_inputs"] = data["surface_samples_inputs"] * scale_inputs
    with precision.storage(mode):
        loss = compute_l2_error(run_forward(model, cfg, d), data["space_samples_tgt"])
        loss.backward()
    torch.cuda.synchronize()
    return float(loss.detach()), {k: p.grad.detach().double() for k, p in model.named_parameters() if p.grad is not None}


def _group_cosines(g_ref, g, depth):
    groups = {}
    for k in g_ref:
        if float(g_ref[k].norm()) > 1e-6:
            c = float((g_ref[k] * g[k]).sum() / (g_ref[k].norm() * g[k].norm() + 1e-30))
            groups.setdefault(".".join(k.split(".")[:depth]), []).append(c)
    return {name: float(np.median(v)) for name, v in groups.items()}


def test_config3_full_size_b32_bf16_step_against_the_fp32_step():
    """BASELINE config 3 at its full size (32 shapes, 2048 surface + 8192 query points, bf16 storage): one train step next
    to the fp32 step of the same product on the same inputs and weights -- loss and the direction of every parameter
    gradient.  No reference bar exists for bf16 (the reference is fp32 only): the numbers are printed and bounded loosely.
    (a) One TDNet (the building block): every module's gradients keep a cosine >= 0.98 to fp32.
    (b) FlowArbitrary feeds the first network's OUTPUT POINTS to the second network's farthest-point sampling and kNN: index
    selection is discontinuous, and with untrained procedural weights the gradients of the upstream modules are chaotic in
    fp32 already -- measured here by an fp32 step on inputs scaled by 1 + 2^-8 (one bf16 ulp): the bf16 step is held to the
    loss, to the last module (downstream of every index), and to that fp32 chaos baseline elsewhere."""
    data = to_dev(synth.make_batch(3232, 32, 2048, 8192), DEV)
    cfg = model_cfg("forward", [2048, 500, 100])
    (l32, g32), (l16, g16) = _step_grads(cfg, data, "f32"), _step_grads(cfg, data, "bf16")
    cos = _group_cosines(g32, g16, 2)
    print(f"\nforward TDNet, B=32: loss fp32 {l32:.6f} bf16 {l16:.6f}; gradient cosine to fp32 per module: min {min(cos.values()):.4f}")
    assert set(g32) == set(g16) and all(bool(torch.isfinite(g).all()) for g in g16.values())
    assert abs(l16 - l32) <= 0.02 * l32
    assert min(cos.values()) >= 0.98, cos

    cfg = model_cfg("arbitrary", [2048, 500, 100])
    (l32, g32), (l16, g16) = _step_grads(cfg, data, "f32"), _step_grads(cfg, data, "bf16")
    _, gpert = _step_grads(cfg, data, "f32", scale_inputs=1.0 + 2.0 ** -8)
    cos, chaos = _group_cosines(g32, g16, 2), _group_cosines(g32, gpert, 2)
    print(f"FlowArbitrary, B=32: loss fp32 {l32:.6f} bf16 {l16:.6f} (rel {abs(l16 - l32) / l32:.2e}); gradient cosine to fp32 "
          f"{ {k: round(v, 3) for k, v in cos.items()} }; fp32 on inputs x (1 + 2^-8): { {k: round(v, 3) for k, v in chaos.items()} }")
    assert set(g32) == set(g16) and all(bool(torch.isfinite(g).all()) for g in g16.values())
    assert abs(l16 - l32) <= 0.05 * l32
    assert cos["model_deform.decoder"] >= 0.95, cos
    for name in cos:
        assert cos[name] >= chaos[name] - 0.5, (name, cos[name], chaos[name])


def test_bf16_storage_against_the_full_size_reference_fixtures():
    """The accuracy cost of bf16 storage measured against the REFERENCE at full point counts (not against this repo's own
    fp32 step): eval L2 and train-step loss of the bf16-storage product vs tests/golden/full_forward.npz (forward.yaml, B = 1)
    and tests/golden/full_arbitrary.npz (arbitrary.yaml, B = 2), both produced by the imported reference on CPU.  No bar
    exists for bf16 in the reference; the numbers are printed (bench.py --dtype bf16 carries them as `parity_l2_vs_fp32`)
    and bounded loosely."""
    from nsdp_amd import precision
    from nsdp_amd.model import optimizer_factory
    # Measured (profiles/r4_bf16_bisect.txt): forward.yaml 8.7e-3 -- twice what inputs x (1 + 2^-8) do to the fp32 model
    # (4.4e-3); arbitrary.yaml 1.3e-1 = network 1's ~9e-3 output error amplified by network 2's FPS / kNN on those points, the
    # same as the fp32 model's response to bf16-rounded input coordinates (1.4e-1); with network 1 in fp32 storage 1.0e-2.
    # The train-step loss of the all-bf16 FlowArbitrary is as ill-conditioned as its eval output: the SAME arithmetic with the batch
    # mean rounded differently (one-launch BatchNorm kernels against the three-launch forms, NSDP_BN_SLAB=1 / 0) gives 0.1520 /
    # 0.1455 against the reference's 0.1443 -- a 4.5 % swing from a 1e-7 perturbation, so that case is held to 8 %; with network
    # 1 in fp32 storage the same switch moves the loss by 1e-3 and the bar stays at 5 %.
    for name, mtype, l2_bound, net1_f32, loss_bound in (("full_forward", "forward", 2e-2, False, 0.05),
                                                        ("full_arbitrary", "arbitrary", 2.5e-1, False, 0.08),
                                                        ("full_arbitrary", "arbitrary", 2.5e-2, True, 0.05)):
        fx, cfg, seed, data = fixture_setup(name, mtype)
        model, train_fn, _ = build_product(cfg, seed, DEV)
        s = int(fx["meta_eval_stride"]) if "meta_eval_stride" in fx else 1
        precision.set_canonicalize_f32(net1_f32)
        with precision.storage(BF):
            model.eval()
            with torch.no_grad():
                out = run_forward(model, cfg, to_dev(data, DEV)).float().cpu().numpy()
            model.train()
            snap = snapshot_model(model)
            _, opt = optimizer_factory({"optimizer": "Adam", "lr": 5e-4}, model.parameters())
            loss = train_fn(model, opt, to_dev(data, DEV), cfg)
            loss_alt = None
            if loss_bound > 0.05:
                # the ill-conditioned case: the variant that lands next to the reference (three-launch BatchNorm: the batch mean
                # summed in another order) keeps the ORIGINAL 5 % bar, so that the 8 % on the default cannot hide a regression
                restore_model(model, snap)
                _, opt2 = optimizer_factory({"optimizer": "Adam", "lr": 5e-4}, model.parameters())
                with batchnorm_three_launch():
                    loss_alt = train_fn(model, opt2, to_dev(data, DEV), cfg)
        precision.set_canonicalize_f32(False)
        l2 = l2_err(out[:, ::s], fx["eval_out"])
        ref_loss = float(fx["train_loss"])
        print(f"\nbf16 storage{' (network 1 in fp32)' if net1_f32 else ''} vs the reference, {name}: eval L2 {l2:.2e}, "
              f"train loss {loss:.6f} (reference {ref_loss:.6f}, rel {abs(loss - ref_loss) / ref_loss:.2e})")
        assert l2 <= l2_bound, (name, l2)
        assert abs(loss - ref_loss) <= loss_bound * ref_loss, (name, loss, ref_loss)
        if loss_alt is not None:
            print(f"   three-launch BatchNorm: train loss {loss_alt:.6f} (rel {abs(loss_alt - ref_loss) / ref_loss:.2e})")
            assert abs(loss_alt - ref_loss) <= 0.05 * ref_loss, (name, loss_alt, ref_loss)
