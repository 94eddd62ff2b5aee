"""G16 layout ([M / 16][C / 4][16 rows][4 floats]) of the tensors between two dense layers: the bf16x3 GEMM and weight-gradient
kernels over G16 operands must produce, bit for bit, what the row-major kernels produce on the same values (the arithmetic is
the same element for element; only addresses change), and a model that keeps its hidden tensors in G16 must be bit-identical
to one that keeps them row-major -- outputs, every gradient, two steps.  Through the C ABI (nsdp_linear_bf16x3_g16_f32,
nsdp_linear_wgrad_bf16x3_g16_f32, nsdp_layout_g16_f32).  Reference tensors: model/encoder/blocks.py:86-124, decoder/blocks.py:30-142."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _rand(g, *shape, scale=1.0):
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


def _bits_ref(h):
    """ReLU bits of h [M, C] with torch ops (the definition): [M / 16][ceil(C / 32)][64] bytes, byte (group, kb, 4 * row + g) =
    bits 0-3 [h > 0] at channels 32 kb + 4 g .. + 3, bits 4-7 at 32 kb + 16 + 4 g .. + 3."""
    M, C = h.shape
    KB = (C + 31) // 32
    pos = torch.zeros(M, KB * 32, dtype=torch.bool, device=h.device)
    pos[:, :C] = h > 0
    p = pos.reshape(M // 16, 16, KB, 2, 4, 4)                 # group, row, kb, half, g, c
    w = torch.tensor([1, 2, 4, 8], device=h.device, dtype=torch.int32)
    nib = (p.to(torch.int32) * w).sum(-1)                     # group, row, kb, half, g
    byte = nib[:, :, :, 0, :] + 16 * nib[:, :, :, 1, :]       # group, row, kb, g
    return byte.permute(0, 2, 1, 3).contiguous().reshape(-1).to(torch.uint8)


def _g16_ref(t):
    """Row-major [M, C] -> G16 with torch ops (the definition)."""
    M, C = t.shape
    return t.reshape(M // 16, 16, C // 4, 4).permute(0, 2, 1, 3).contiguous().reshape(M, C)


@pytest.mark.parametrize("M,C", [(16, 4), (48, 200), (4096, 256), (65536 + 16, 120)])
def test_layout_round_trip_and_definition(M, C):
    from nsdp_amd import hip_linear
    g = torch.Generator(device="cpu").manual_seed(M + C)
    t = _rand(g, M, C)
    tg = hip_linear.to_g16(t)
    assert torch.equal(tg, _g16_ref(t))
    assert torch.equal(hip_linear.to_g16(tg, back=True), t)


# (M, K, N): 8-tile resident-weight and two-workgroup forms, 13-tile 8-wave and register-path forms, 16-tile 2- and 3-row-tile forms;
# M a multiple of 16 only (ragged last workgroup tile), several tiles per persistent workgroup
GEMM_SHAPES = [(65536, 128, 128), (32768 + 16, 120, 120), (65536 + 48, 200, 128), (262144, 200, 200), (256 * 256 * 3 + 16, 200, 200),
               (51200, 256, 256), (256 * 192 * 2 + 32, 256, 256), (40000, 256, 200)]


@pytest.mark.parametrize("M,K,N", GEMM_SHAPES)
@pytest.mark.parametrize("form", ["y_plain", "y_relu_in", "x_plain", "x_masked"])
def test_linear_g16_is_bit_equal_to_row_major(M, K, N, form):
    from nsdp_amd import hip_linear as hl
    L = hl.lib()
    g = torch.Generator(device="cpu").manual_seed(M + 7 * K + 31 * N)
    x, w, b = _rand(g, M, K), _rand(g, N, K, scale=K ** -0.5), _rand(g, N)
    wp = hl.pack_weight_x3(w)[0]
    if form == "y_plain":        # first layer forward / second layer dX: bias + output ReLU
        if not L.nsdp_linear_bf16x3_g16_supported(hl._ll(M), N, K, 2, 0, 0):
            pytest.skip("form not instantiated")
        ref = hl._fwd_x3(x, wp, N, b, None, None, None, False, True)
        got = hl._fwd_x3_g16(x, wp, N, b, None, None, None, False, True, hl.LAY_Y)
        assert torch.equal(hl.to_g16(got, back=True), ref)
        bits = hl.relu_bits(M, N, DEV).fill_(0xAA)             # ... and the ReLU bits the same launch writes on request
        got = hl._fwd_x3_g16(x, wp, N, b, None, None, None, False, True, hl.LAY_Y, bits_out=bits)
        assert torch.equal(hl.to_g16(got, back=True), ref)
        assert torch.equal(bits, _bits_ref(ref))
    elif form == "y_relu_in":    # ResnetBlockFC's fc_0
        if not L.nsdp_linear_bf16x3_g16_supported(hl._ll(M), N, K, 2, 0, 1):
            pytest.skip("form not instantiated")
        ref = hl._fwd_x3(x, wp, N, b, None, None, None, True, True)
        got = hl._fwd_x3_g16(x, wp, N, b, None, None, None, True, True, hl.LAY_Y)
        assert torch.equal(hl.to_g16(got, back=True), ref)
    elif form == "x_plain":      # second layer forward: X in G16, row-major residual and output
        if not L.nsdp_linear_bf16x3_g16_supported(hl._ll(M), N, K, 1, 0, 0):
            pytest.skip("form not instantiated")
        res = _rand(g, M, N)
        ref = hl._fwd_x3(x, wp, N, b, res, None, None, False, False)
        got = hl._fwd_x3_g16(hl.to_g16(x), wp, N, b, res, None, None, False, False, hl.LAY_X)
        assert torch.equal(got, ref)
    else:                        # first layer dX: dY and the ReLU mask in G16; row-major residual, out_mask, addend
        if not L.nsdp_linear_bf16x3_g16_supported(hl._ll(M), N, K, 1, 1, 0):
            pytest.skip("form not instantiated")
        mask, res = torch.relu(_rand(g, M, K)), _rand(g, M, N)
        ref = hl._fwd_x3(x, wp, N, None, res, mask, None, False, False)
        got = hl._fwd_x3_g16(hl.to_g16(x), wp, N, None, res, hl.to_g16(mask), None, False, False, hl.LAY_X)
        assert torch.equal(got, ref)
        omask, add = torch.relu(_rand(g, M, N)), _rand(g, M, N)
        ref = hl._fwd_x3(x, wp, N, None, None, mask, omask, False, False, addend=add)
        got = hl._fwd_x3_g16(hl.to_g16(x), wp, N, None, None, hl.to_g16(mask), omask, False, False, hl.LAY_X, addend=add)
        assert torch.equal(got, ref)
        # ... and the same two with the mask as ReLU bits (one bit per element, the layout of csrc/x3_kernel.h)
        bits = _bits_ref(mask)
        got = hl._fwd_x3_g16(hl.to_g16(x), wp, N, None, res, None, None, False, False, hl.LAY_X, mask_bits=bits)
        assert torch.equal(got, hl._fwd_x3(x, wp, N, None, res, mask, None, False, False))
        got = hl._fwd_x3_g16(hl.to_g16(x), wp, N, None, None, None, omask, False, False, hl.LAY_X, addend=add, mask_bits=bits)
        assert torch.equal(got, ref)


@pytest.mark.parametrize("M,N,K", [(65536, 128, 128), (32768 + 32, 120, 120), (262144 + 64, 200, 200), (51200, 256, 256), (100000 * 32 // 32 * 32, 200, 200)])
@pytest.mark.parametrize("layout,masked", [(1, True), (1, "bits"), (1, False), (2, False)])
def test_wgrad_g16_is_bit_equal_to_row_major(M, N, K, layout, masked):
    from nsdp_amd import hip_linear as hl
    if not hl.lib().nsdp_linear_wgrad_bf16x3_g16_supported(hl._ll(M), N, K, layout, int(bool(masked))):
        pytest.skip("form not instantiated")
    g = torch.Generator(device="cpu").manual_seed(M + 3 * N + 17 * K + layout)
    dy, x = _rand(g, M, N), _rand(g, M, K)
    mask = torch.relu(_rand(g, M, N)) if masked else None
    dw0, db0 = hl._wgrad_x3(dy, x, mask, True, True)
    fn = hl._wgrad_g16_fn(layout, _bits_ref(mask) if masked == "bits" else None)
    if layout == 1:
        dw1, db1 = fn(hl.to_g16(dy), x, hl.to_g16(mask) if masked else None, True, True)
    else:
        dw1, db1 = fn(dy, hl.to_g16(x), None, True, True)
    assert torch.equal(dw1, dw0) and torch.equal(db1, db0)
    # accumulate into existing buffers
    acc_w, acc_b = torch.ones_like(dw0), torch.ones_like(db0)
    ref_w, ref_b = hl._wgrad_x3(dy, x, mask, True, True, out=(acc_w.clone(), acc_b.clone()))
    if layout == 1:
        got_w, got_b = fn(hl.to_g16(dy), x, hl.to_g16(mask) if masked else None, True, True, out=(acc_w.clone(), acc_b.clone()))
    else:
        got_w, got_b = fn(dy, hl.to_g16(x), None, True, True, out=(acc_w.clone(), acc_b.clone()))
    assert torch.equal(got_w, ref_w) and torch.equal(got_b, ref_b)


def _mlp_step(g16, M, K, H, N, resnet, seed):
    """Two SGD-free steps of y = pair(x): forward, backward of sum(y * t), gradients of x and the four parameters."""
    import torch.nn as nn
    from nsdp_amd import hip_linear
    from nsdp_amd.model import ops
    from nsdp_amd.model.decoder.blocks import ResnetBlockFC
    was = hip_linear.G16
    hip_linear.G16 = g16
    try:
        torch.manual_seed(seed)
        if resnet:
            blk = ResnetBlockFC(K).to(DEV)
            nn.init.normal_(blk.fc_1.weight, std=K ** -0.5)
            params = list(blk.parameters())
            f = blk
        else:
            seq = nn.Sequential(nn.Linear(K, H), nn.ReLU(), nn.Linear(H, N)).to(DEV)
            params = list(seq.parameters())
            f = lambda x: ops.mlp2(x, seq)
        g = torch.Generator(device="cpu").manual_seed(seed)
        x = _rand(g, M, K).requires_grad_(True)
        t = _rand(g, M, N if not resnet else K)
        outs = []
        for _ in range(2):
            for p in params:
                p.grad = None
            x.grad = None
            y = f(x)
            (y * t).sum().backward()
            torch.cuda.synchronize()
            outs.append([y.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in params])
        return outs
    finally:
        hip_linear.G16 = was


@pytest.mark.parametrize("M,K,H,N,resnet", [(65536, 200, 200, 200, False), (51200, 256, 256, 256, False), (65536, 128, 128, 128, True),
                                            (32768 + 32, 120, 120, 120, False)])
def test_layer_pair_with_a_g16_hidden_tensor_is_bit_equal(M, K, H, N, resnet):
    import ctypes
    from nsdp_amd import hip_linear
    from nsdp_amd.model import ops
    if not hip_linear.G16 or ops.PAIR_MASK:
        pytest.skip("NSDP_G16=0 / NSDP_PAIR_MASK=1: the pair keeps its hidden tensor row-major")
    L = hip_linear.lib()
    assert hip_linear.g16_pair_ok(M, K, H, N, relu_in0=resnet, train=True), "the pair should take the G16 form at this shape"
    L.nsdp_trace_enable(1)
    a = _mlp_step(True, M, K, H, N, resnet, 5)
    L.nsdp_trace_enable(0)
    n = L.nsdp_trace_read(None, 0)
    buf = ctypes.create_string_buffer(n)
    L.nsdp_trace_read(buf, n)
    trace = buf.value.decode()
    assert "g16:y" in trace and "g16:x" in trace and "g16:dy" in trace, trace[:2000]      # the forms ran
    if hip_linear.G16_BITS:
        assert ",bits,notail> g16:dy" in trace, trace[:2000]                               # ... with the mask as ReLU bits
    b = _mlp_step(False, M, K, H, N, resnet, 5)
    for sa, sb in zip(a, b):
        for ta, tb in zip(sa, sb):
            assert torch.equal(ta, tb)
