"""bf16x3 split-precision dense kernels (nsdp_linear_bf16x3_f32, nsdp_linear_wgrad_bf16x3_f32) and the fragment-major
weight packs, through the C ABI, against an fp64 reference of the same op.  The bar is the exact-fp32 MFMA kernel's
own error: the 3-way split must not be measurably less accurate (tolerances are stated per test)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _rand(g, *shape, scale=1.0):
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


def _ref64(x, w, b, res, mask, out_mask, relu_in, relu_out):
    xi = x.double()
    if mask is not None:
        xi = xi * (mask > 0)
    if relu_in:
        xi = F.relu(xi)
    y = xi @ w.double().t()
    if b is not None:
        y = y + b.double()
    if res is not None:
        y = y + res.double()
    if relu_out:
        y = F.relu(y)
    if out_mask is not None:
        y = y * (out_mask > 0)
    return y


X3_CASES = [  # (M, K, N, bias, residual, mask, out_mask, relu_in, relu_out)
    (256, 64, 64, True, False, False, False, False, False),          # one workgroup tile, two k blocks
    (1000, 200, 200, True, False, False, False, False, True),        # ragged rows, ragged k block (200 = 6.25 x 32)
    (4099, 120, 120, False, True, False, False, False, False),       # residual as accumulator init, N % 16 != 0
    (70001, 128, 256, True, False, True, False, False, False),       # mask prologue, N = 256 (16 tiles), many tiles per workgroup
    (33000, 256, 128, True, False, False, True, False, False),       # out_mask epilogue
    (9000, 200, 128, True, True, False, False, True, True),          # relu on both sides + residual
    (300, 36, 200, True, False, False, False, False, False),         # K just above one k block
    # several row blocks per persistent workgroup (the prefetch of the next block's first k blocks under the epilogue, the
    # ragged last block), one case per width class / kernel form
    (256 * 256 * 3 + 77, 200, 200, True, False, False, False, False, True),      # 13 n tiles, 8 waves
    (256 * 256 * 3 + 77, 200, 200, False, False, False, False, True, False),     # no bias, ReLU prologue
    (256 * 192 * 3 + 5, 256, 256, True, False, False, False, False, False),      # 16 n tiles, 4 waves x 3 row tiles
    (256 * 256 * 3 + 31, 96, 104, True, False, False, False, False, True),       # resident-weight form (<= 128 wide)
    # the masked (register-path) forms with several row blocks per workgroup, and the output mask through the staged epilogue
    (128 * 512 * 2 + 50, 128, 128, False, False, True, True, False, False),      # dX of a relu_in layer: mask + out_mask, 8 tiles
    (192 * 256 * 2 + 77, 200, 200, False, True, True, False, False, False),      # mask + residual, 13 tiles, 3 row tiles per wave
    (128 * 256 * 3 + 9, 256, 256, True, False, True, True, False, False),        # 16 tiles, mask + out_mask
    (256 * 256 * 2 + 33, 200, 200, False, False, False, True, False, False),     # out_mask alone, 8-wave form
]


@pytest.mark.parametrize("M,K,N,bias,res,mask,omask,relu_in,relu_out", X3_CASES)
def test_linear_bf16x3_matches_fp64(M, K, N, bias, res, mask, omask, relu_in, relu_out):
    from nsdp_amd import hip_linear
    g = torch.Generator(device="cpu").manual_seed(M + 13 * K + 101 * N)
    x, w = _rand(g, M, K), _rand(g, N, K, scale=K ** -0.5)
    b = _rand(g, N) if bias else None
    r = _rand(g, M, N) if res else None
    m = _rand(g, M, K) if mask else None
    o = _rand(g, M, N) if omask else None
    ref = _ref64(x, w, b, r, m, o, relu_in, relu_out)
    y3 = hip_linear._fwd_x3(x, hip_linear.pack_weight_x3(w)[0], N, b, r, m, o, relu_in, relu_out)
    y32 = hip_linear._fwd_wp(x, hip_linear.pack_weight(w)[0], N, b, r, m, o, relu_in, relu_out)
    scale = float(ref.abs().max())
    e3, e32 = float((y3.double() - ref).abs().max()) / scale, float((y32.double() - ref).abs().max()) / scale
    # fp32 rounding level: 2^-24 = 6e-8 per operation, sqrt(K)-ish growth over the reduction
    assert e3 <= 1.5e-6, (e3, e32)
    assert e3 <= 2.0 * e32 + 2e-7, (e3, e32)      # not measurably worse than the exact-fp32 MFMA chain


def test_linear_bf16x3_transposed_pack_is_dx():
    """dX = dY @ W through the pack of W^T (backward operand), ragged N as the reduction dimension."""
    from nsdp_amd import hip_linear
    g = torch.Generator(device="cpu").manual_seed(5)
    M, N, K = 5000, 200, 120
    dy, w = _rand(g, M, N), _rand(g, N, K)
    wpt = hip_linear.pack_weight_x3(w, False, True)[1]
    dx = hip_linear._fwd_x3(dy, wpt, K, None, None, None, None, False, False)
    ref = dy.double() @ w.double()
    assert float((dx.double() - ref).abs().max()) / float(ref.abs().max()) <= 1.5e-6


def test_fragment_major_pack_kernel_is_exact_against_rowmajor_kernel():
    """nsdp_linear_wp_f32 (packed W) must be bit-identical to nsdp_linear_f32 (row-major W): same MFMA chain."""
    from nsdp_amd import hip_linear
    g = torch.Generator(device="cpu").manual_seed(9)
    for (M, K, N) in [(777, 200, 200), (64, 4, 120), (5000, 128, 3), (33, 256, 256)]:
        x, w, b = _rand(g, M, K), _rand(g, N, K), _rand(g, N)
        wp, wpt = hip_linear.pack_weight(w, True, True)
        assert torch.equal(hip_linear._fwd_wp(x, wp, N, b, None, None, None, False, True),
                           hip_linear._fwd(x, w, b, None, None, None, False, True))
        if N % 4 == 0:
            dy = _rand(g, M, N)
            assert torch.equal(hip_linear._fwd_wp(dy, wpt, K, None, None, None, None, False, False),
                               hip_linear._fwd(dy, w.t().contiguous(), None, None, None, None, False, False))


WG_CASES = [  # (M, N, K, mask, relu_x)
    (2048, 200, 200, False, False),
    (4099, 120, 128, True, True),        # ragged last 32-row block, both prologues
    (70000, 128, 200, True, False),      # mixed tile classes (8 x 13)
    (50000, 200, 120, False, True),      # (13 x 8)
    (33333, 64, 64, False, False),
    (5000, 256, 256, True, False),       # N = 256 (16 tiles), columns of X split over grid.y
    (3200, 200, 256, False, False),      # K = 256 with a 13-tile dY
    (2500, 256, 200, False, True),       # 16 + 13 tiles would not fit in LDS: K split 128 + 72
    (4096, 120, 128, True, True),        # M % 32 == 0: the variants without row clamps / row masks
    (65536, 200, 120, True, True),
    (32768, 256, 200, False, True),
]


@pytest.mark.parametrize("M,N,K,mask,relu_x", WG_CASES)
def test_wgrad_bf16x3_matches_fp64(M, N, K, mask, relu_x):
    from nsdp_amd import hip_linear
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    dy, x = _rand(g, M, N), _rand(g, M, K)
    m = _rand(g, M, N) if mask else None
    dw, db = hip_linear._wgrad_x3(dy, x, m, relu_x, True)
    dyp = dy.double() * (m > 0) if mask else dy.double()
    xp = F.relu(x.double()) if relu_x else x.double()
    ref_w, ref_b = dyp.t() @ xp, dyp.sum(0)
    # rows are the reduction dimension: fp32 accumulation over M terms (partials per workgroup, then a fixed-order sum)
    assert float((dw.double() - ref_w).abs().max()) / float(ref_w.abs().max()) <= 3e-6
    assert float((db.double() - ref_b).abs().max()) / float(ref_b.abs().max()) <= 3e-6
    dw2, db2 = hip_linear._wgrad_x3(dy, x, m, relu_x, True)
    assert torch.equal(dw, dw2) and torch.equal(db, db2)          # deterministic


def test_bf16x3_rejects_shapes_outside_its_contract():
    from nsdp_amd import _lib, hip_linear
    x = torch.randn(128, 32, device=DEV)
    w = torch.randn(64, 32, device=DEV)
    with pytest.raises(_lib.NsdpHipError):         # K must span two k blocks
        hip_linear._fwd_x3(x, hip_linear.pack_weight_x3(w)[0], 64, None, None, None, None, False, False)
    with pytest.raises(_lib.NsdpHipError):         # K = 16: a single tile is below the kernel's range
        hip_linear._wgrad_x3(torch.randn(4096, 256, device=DEV), torch.randn(4096, 16, device=DEV), None, False, True)


def test_autograd_routes_large_layers_through_bf16x3_and_matches_fp32_path():
    """End to end through hip_linear.linear: a layer large enough for the split path, against the exact path."""
    from nsdp_amd import hip_linear
    g = torch.Generator(device="cpu").manual_seed(3)
    M, K, N = 40000, 200, 200
    x0, w0, b0 = _rand(g, M, K), _rand(g, N, K, scale=K ** -0.5), _rand(g, N)
    go = _rand(g, M, N)
    outs = []
    for use in (True, False):
        hip_linear._USE_X3 = use
        try:
            x, w, b = (t.clone().requires_grad_(True) for t in (x0, w0, b0))
            y = hip_linear.linear(x, w, b)      # (no ReLU: a sign flip of y ~ 0 between the paths would change dX)
            outs.append((y.detach(),) + torch.autograd.grad(y, (x, w, b), go))
        finally:
            hip_linear._USE_X3 = True
    for a, e in zip(*outs):
        s = float(e.abs().max())
        assert float((a - e).abs().max()) <= 4e-6 * s


@pytest.mark.parametrize("M,K,N,res,relu_in,relu_out", [(524288 + 129 * 7, 200, 200, False, False, True),
                                                        (600001, 64, 144, True, True, False)])
def test_linear_bf16x3_antiphase_variant_matches_fp64(M, K, N, res, relu_in, relu_out):
    """The experimental anti-phase form of the 8-wave kernel (nsdp_debug_set(6, 128): two 4-wave groups half a tile apart
    on one rotating weight stream, csrc/gemm_bf16x3.hip) computes the same layer: fp32-level error against fp64 on sampled
    rows (ragged last tile included), and within rounding of the standard form (the k blocks are summed in rotated order)."""
    from nsdp_amd import _lib, hip_linear
    g = torch.Generator(device="cpu").manual_seed(M + K)
    x, w, b = _rand(g, M, K), _rand(g, N, K, scale=K ** -0.5), _rand(g, N)
    r = _rand(g, M, N) if res else None
    wp = hip_linear.pack_weight_x3(w)[0]
    y_std = hip_linear._fwd_x3(x, wp, N, b, r, None, None, relu_in, relu_out)
    _lib.lib().nsdp_debug_set(6, 128)
    try:
        y_ap = hip_linear._fwd_x3(x, wp, N, b, r, None, None, relu_in, relu_out)
        torch.cuda.synchronize()
    finally:
        _lib.lib().nsdp_debug_set(6, 0)
    idx = torch.cat([torch.arange(0, 256, device=DEV), torch.arange(M - 256, M, device=DEV),
                     torch.randint(0, M, (4096,), device=DEV)])
    ref = _ref64(x[idx], w, b, None if r is None else r[idx], None, None, relu_in, relu_out)
    scale = float(ref.abs().max())
    assert float((y_ap[idx].double() - ref).abs().max()) / scale <= 1.5e-6
    assert float((y_ap - y_std).abs().max()) / scale <= 3e-6


@pytest.mark.parametrize("B,n,k,nsrc,K,N,per_shape,bias,relu_in", [
    (2, 1100, 16, 300, 200, 200, False, True, False),      # per-point rows (set abstraction / transformer block), 13 n tiles
    (3, 4099, 7, 100, 200, 200, True, True, False),        # one query per shape (decoder), ragged row count
    (2, 9000, 7, 100, 256, 256, True, False, False),       # ... 16 n tiles
    (4, 3000, 7, 100, 128, 128, True, True, False),        # ... resident-weight form (direct epilogue)
    (2, 1024, 16, 512, 128, 128, False, False, False),     # resident-weight form (direct epilogue), no bias
    (4, 2048, 16, 2048, 120, 120, False, True, False),     # ... ragged last n tile (the set-abstraction layers)
    (1, 70000, 10, 2048, 256, 256, False, True, False),    # 16 n tiles, several row blocks per workgroup
])
def test_linear_bf16x3_gathered_addend(B, n, k, nsrc, K, N, per_shape, bias, relu_in):
    """nsdp_linear_bf16x3_gather_f32: y[r] = x[r] @ W^T + b + (gq[r / g_div] - gk[(r / rows_per_shape) * nsrc + gidx[r]]) -- the
    position-encoding MLP's last layer producing u = q_i - k_j + pos directly.  The addend joins in the epilogue, so the
    result carries the plain kernel's error plus one rounding of the sum (an accumulator STARTED from q - k would round
    every small product at the difference's ulp)."""
    from nsdp_amd import hip_linear
    g = torch.Generator(device="cpu").manual_seed(B + n + 7 * k + N)
    M = B * n * k
    x, w = _rand(g, M, K, scale=0.3), _rand(g, N, K, scale=K ** -0.5)       # pos of a fraction of a unit under q - k of a few
    b = _rand(g, N) if bias else None
    gq = _rand(g, B if per_shape else B * n, N, scale=2.0)
    gk = _rand(g, B * nsrc, N, scale=2.0)
    gidx = torch.randint(0, nsrc, (M,), generator=g).int().to(DEV)
    assert hip_linear.gather_init_ok(M, N, K)
    wp = hip_linear.pack_weight_x3(w)[0]
    rows = torch.arange(M, device=DEV)
    if per_shape:      # the one-table form: q_b - k_bj prepared by the caller
        table = (gq.view(B, 1, N) - gk.view(B, nsrc, N)).reshape(-1, N)
        y = hip_linear._fwd_x3_gather(x, wp, N, b, (None, 1, table, gidx, n * k, nsrc), relu_in, False)
        add = table[(rows // (n * k)) * nsrc + gidx.long()]
    else:
        y = hip_linear._fwd_x3_gather(x, wp, N, b, (gq, k, gk, gidx, n * k, nsrc), relu_in, False)
        add = gq[rows // k] - gk[(rows // (n * k)) * nsrc + gidx.long()]      # one fp32 subtraction
    ref = _ref64(x, w, b, add, None, None, relu_in, False)
    plain = hip_linear._fwd_x3(x, wp, N, b, None, None, None, relu_in, False)
    err = float((y.double() - ref).abs().max())
    # fl(plain + add): half an ulp of the sum on top of the plain kernel's own error
    sum_ulp = float(ref.abs().max()) * 2.0 ** -23
    assert err <= float((plain.double() - _ref64(x, w, b, None, None, None, relu_in, False)).abs().max()) + 0.5 * sum_ulp + 1e-9
    assert torch.equal(y, plain + add)      # exactly that: the epilogue adds the two fp32 values


@pytest.mark.parametrize("M,K,N,bias", [(40000, 128, 128, False), (33000, 120, 120, True), (70001, 200, 200, True),
                                        (50000, 256, 256, False)])
def test_linear_bf16x3_signed_residual(M, K, N, bias):
    """nsdp_linear_bf16x3_signed_f32 with sign -1 == the plain kernel fed the negated residual, bit for bit (the projections
    "minus a table" of a set abstraction's second attention); through hip_linear.linear the residual's gradient is -dy."""
    from nsdp_amd import hip_linear
    g = torch.Generator(device="cpu").manual_seed(M + N)
    x, w, r = _rand(g, M, K), _rand(g, N, K, scale=K ** -0.5), _rand(g, M, N)
    b = _rand(g, N) if bias else None
    wp = hip_linear.pack_weight_x3(w)[0]
    y = hip_linear._fwd_x3(x, wp, N, b, r, None, None, False, False, res_sign=-1.0)
    assert torch.equal(y, hip_linear._fwd_x3(x, wp, N, b, -r, None, None, False, False))
    ref = _ref64(x, w, b, -r, None, None, False, False)
    assert float((y.double() - ref).abs().max()) / float(ref.abs().max()) <= 1.5e-6
    xr, rr = x[:4096].clone().requires_grad_(True), r[:4096].clone().requires_grad_(True)      # (small M: the exact-fp32 kernel's route)
    out = hip_linear.linear(xr, w, b, residual=rr, residual_sign=-1.0)
    go = _rand(g, 4096, N)
    dx, dr = torch.autograd.grad(out, [xr, rr], go)
    assert torch.equal(dr, -go)
    assert float((out.double() - _ref64(x[:4096], w, b, -r[:4096], None, None, False, False)).abs().max()) <= 2e-5
    assert float((dx.double() - go.double() @ w.double()).abs().max()) <= 2e-5 * float(go.abs().max()) * K ** 0.5


@pytest.mark.parametrize("M,K,N,res", [(128 * 512 * 2 + 50, 128, 128, False), (192 * 256 * 2 + 77, 200, 200, True),
                                       (40000, 256, 256, False), (33001, 120, 120, False)])
def test_linear_bf16x3_addend_after_the_output_mask(M, K, N, res):
    """nsdp_linear_bf16x3_addend_f32: gate(out_mask) of the masked GEMM, THEN + addend -- bit for bit the masked kernel's output
    plus the addend (one fp32 add), i.e. dX of a residual block's first layer with the skip gradient joined in the epilogue."""
    from nsdp_amd import hip_linear
    g = torch.Generator(device="cpu").manual_seed(M + 3 * N)
    x, w = _rand(g, M, K), _rand(g, N, K, scale=K ** -0.5)
    m, o, a = _rand(g, M, K), _rand(g, M, N), _rand(g, M, N)
    r = _rand(g, M, N) if res else None
    wp = hip_linear.pack_weight_x3(w)[0]
    y = hip_linear._fwd_x3(x, wp, N, None, r, m, o, False, False, addend=a)
    plain = hip_linear._fwd_x3(x, wp, N, None, r, m, o, False, False)
    assert torch.equal(y, plain + a)
    ref = _ref64(x, w, None, r, m, o, False, False) + a.double()
    assert float((y.double() - ref).abs().max()) / float(ref.abs().max()) <= 1.5e-6
