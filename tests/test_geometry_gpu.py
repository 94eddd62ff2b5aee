"""GPU parity of the geometry / pointnet2 operators against the CPU oracle (bit-exact indices)."""
import os

import numpy as np
import pytest
import torch

from nsdp_amd import synth
from oracle import pointnet2_ref as ref

pytestmark = pytest.mark.gpu


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _cloud(seed, b, n, kind="uniform"):
    xyz = synth.uniform(seed, f"cloud{kind}", (b, n, 3), -0.5, 0.5)
    if kind == "origin":       # many points inside the mag <= 1e-3 ball (skipped by the kernel)
        xyz[:, ::7] *= 0.03
    elif kind == "dupes":      # exact duplicates -> exact distance ties (fp16-stored real data has them)
        xyz[:, 1::2] = xyz[:, 0::2][:, : xyz[:, 1::2].shape[1]]
    elif kind == "grid":       # lattice: massive ties everywhere
        g = np.stack(np.meshgrid(*[np.arange(16)] * 3, indexing="ij"), -1).reshape(-1, 3)
        xyz = np.tile(((g[:n] - 7.5) / 16.0).astype(np.float32)[None], (b, 1, 1))
    elif kind == "allorigin":
        xyz = np.zeros((b, n, 3), np.float32)
    return np.ascontiguousarray(xyz, dtype=np.float32)


@pytest.mark.parametrize("n,m", [(1, 1), (3, 3), (64, 16), (100, 100), (256, 64), (500, 100), (513, 77),
                                 (2048, 500), (3000, 300), (5000, 500), (8192, 64), (10000, 40)])
def test_fps_matches_oracle(n, m):
    from nsdp_amd import pointnet2_utils as pu
    xyz = _cloud(n * 7 + m, 3, n)
    got = pu.furthest_point_sample(_dev(xyz), m)
    assert got.dtype == torch.int32 and tuple(got.shape) == (3, m)
    np.testing.assert_array_equal(got.cpu().numpy(), ref.furthest_point_sampling(xyz, m))


@pytest.mark.parametrize("kind", ["origin", "dupes", "grid", "allorigin"])
@pytest.mark.parametrize("n,m", [(500, 100), (2048, 500), (4096, 200)])
def test_fps_edge_cases_match_oracle(kind, n, m):
    """Skipped near-origin points, exact ties (tie rule of the reference's block tree), degenerate clouds."""
    from nsdp_amd import pointnet2_utils as pu
    xyz = _cloud(11, 2, n, kind)
    got = pu.furthest_point_sample(_dev(xyz), m).cpu().numpy()
    np.testing.assert_array_equal(got, ref.furthest_point_sampling(xyz, m))


def test_fps_golden_pyramid(golden_dir):
    """FPS pyramid of the golden fixtures (tiny: 256->64->16, full: 2048->500->100)."""
    from nsdp_amd import pointnet2_utils as pu
    for name in ("tiny_forward", "full_forward"):
        fx = np.load(os.path.join(golden_dir, name + ".npz"))
        seed, b, ns, nq = (int(fx[k]) for k in ("meta_seed", "meta_batch", "meta_ns", "meta_nq"))
        npl = [int(x) for x in fx["meta_npl"]]
        xyz0 = _dev(synth.make_batch(seed, b, ns, nq)["surface_samples_inputs"][:, :, :3])
        fps1 = pu.furthest_point_sample(xyz0, npl[1])
        xyz1 = pu.gather_rows(xyz0, fps1)
        fps2 = pu.furthest_point_sample(xyz1, npl[2])
        np.testing.assert_array_equal(fps1.cpu().numpy(), fx["geo/fps1"])
        np.testing.assert_array_equal(fps2.cpu().numpy(), fx["geo/fps2"])


@pytest.mark.parametrize("n,m,k", [(1, 1, 1), (5, 9, 3), (100, 100, 16), (500, 2048, 16), (2048, 2048, 10),
                                   (8192, 100, 7), (300, 5000, 16), (64, 1500, 33), (70, 70, 64)])
def test_knn_matches_oracle(n, m, k):
    from nsdp_amd import pointnet2_utils as pu
    q = _cloud(n + k, 2, n)
    s = q if n == m else _cloud(m + 5 * k, 2, m)
    idx, d2 = pu.knn(_dev(q), _dev(s), k, return_dist=True)
    ridx, rd2 = ref.knn(q, s, k, return_dist=True)
    np.testing.assert_array_equal(idx.cpu().numpy(), ridx)
    assert np.array_equal(d2.cpu().numpy().view(np.uint32), rd2.view(np.uint32))  # bit-exact distances


def test_knn_ties_follow_distance_then_index():
    from nsdp_amd import pointnet2_utils as pu
    s = _cloud(3, 2, 512, "grid")
    idx = pu.knn(_dev(s), _dev(s), 16).cpu().numpy()
    np.testing.assert_array_equal(idx, ref.knn(s, s, 16))


@pytest.mark.parametrize("kind,b,n,m,k", [("uniform", 3, 2048, 2048, 16), ("dupes", 2, 500, 2048, 16), ("grid", 2, 512, 512, 16),
                                          ("uniform", 2, 300, 5000, 16), ("allorigin", 2, 300, 700, 9), ("uniform", 2, 257, 1025, 5)])
def test_knn_deferred_insertion_equals_immediate_insertion(kind, b, n, m, k):
    """knn_split_queue_kernel (candidates appended to per-lane FIFOs, lists updated in rounds) is the default of the
    several-lanes-per-query search; knn_split_kernel (insert at once) stays as its A/B form.  Same candidate order per lane,
    same strict `<` chain: indices and distance bits must be equal on ties, duplicates, ragged tiles and all."""
    from test_model_gpu import _variant_trace
    from nsdp_amd import _lib
    from nsdp_amd import pointnet2_utils as pu
    s = _cloud(11 * n + m, b, m, kind)
    q = s[:, :n].copy() if n <= m else _cloud(n, b, n, kind)
    out = {}
    for mode in (0, 1):
        _lib.lib().nsdp_debug_set(10, mode)
        try:
            with _variant_trace() as names:
                idx, d2 = pu.knn(_dev(q), _dev(s), k, return_dist=True)
                torch.cuda.synchronize()
        finally:
            _lib.lib().nsdp_debug_set(10, int(os.environ.get("NSDP_KNN_QUEUE", "1")))
        assert any(x.startswith("knn_split_queue<" if mode else "knn_split<") for x in names), names
        out[mode] = (idx.cpu().numpy(), d2.cpu().numpy().view(np.uint32))
    np.testing.assert_array_equal(out[0][0], out[1][0])
    np.testing.assert_array_equal(out[0][1], out[1][1])
    ridx = ref.knn(q, s, k)
    if kind != "allorigin":        # (all distances equal: the oracle's order is distance-then-index as well, checked below)
        np.testing.assert_array_equal(out[1][0], ridx)
    else:
        np.testing.assert_array_equal(out[1][0], np.tile(np.arange(k, dtype=np.int32), (b, n, 1)))


def test_knn_golden_sets(golden_dir):
    """kNN sets produced by the reference's own square_distance + argsort (tiny fixture, every site)."""
    from nsdp_amd import pointnet2_utils as pu
    fx = np.load(os.path.join(golden_dir, "tiny_forward.npz"))
    seed, b, ns, nq = (int(fx[k]) for k in ("meta_seed", "meta_batch", "meta_ns", "meta_nq"))
    data = synth.make_batch(seed, b, ns, nq)
    xyz0 = _dev(data["surface_samples_inputs"][:, :, :3])
    xyz1 = pu.gather_rows(xyz0, _dev(fx["geo/fps1"]))
    xyz2 = pu.gather_rows(xyz1, _dev(fx["geo/fps2"]))
    q = _dev(data["space_samples_src"])
    sites = {"begin": (xyz0, xyz0, 10), "tsa0": (xyz1, xyz0, 16), "down0": (xyz1, xyz1, 16),
             "tsa1": (xyz2, xyz1, 16), "down1": (xyz2, xyz2, 16), "dec": (q, xyz2, 7)}
    for name, (a, s, k) in sites.items():
        np.testing.assert_array_equal(pu.knn(a, s, k).cpu().numpy(), fx["geo/knn_" + name], err_msg=name)


def test_gather_and_group_ops_match_oracle():
    from nsdp_amd import pointnet2_utils as pu
    B, C, N, M, NS = 3, 7, 50, 23, 5
    feats = synth.normal(1, "feats", (B, C, N))
    idx = (synth.uniform01(2, "idx", (B, M)) * N).astype(np.int32)
    gidx = (synth.uniform01(3, "gidx", (B, M, NS)) * N).astype(np.int32)
    f = _dev(feats).requires_grad_(True)
    out = pu.gather_operation(f, _dev(idx))
    np.testing.assert_array_equal(out.detach().cpu().numpy(), ref.gather_points(feats, idx))
    go = synth.normal(4, "go", (B, C, M))
    out.backward(_dev(go))
    np.testing.assert_allclose(f.grad.cpu().numpy(), ref.gather_points_grad(go, idx, N), rtol=1e-6, atol=1e-6)
    f.grad = None
    out = pu.grouping_operation(f, _dev(gidx))
    np.testing.assert_array_equal(out.detach().cpu().numpy(), ref.group_points(feats, gidx))
    go = synth.normal(5, "go2", (B, C, M, NS))
    out.backward(_dev(go))
    np.testing.assert_allclose(f.grad.cpu().numpy(), ref.group_points_grad(go, gidx, N), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("B,C,N,M,NS", [(2, 8, 64, 32, 4), (3, 13, 1000, 257, 3), (2, 33, 20000, 500, 16), (1, 5, 40000, 77, 8),
                                          (4, 128, 2048, 500, 16),
                                          # rows of E = M x NS > 32768 entries do not fit LDS: the sliced list kernel
                                          # (cursors into ascending lists; the hot point's list is longer than the list
                                          # build sorts -> its whole-list path), N > 8192 sources, 2 / 4 channels per walk
                                          (2, 6, 8192, 2048, 32), (1, 9, 16384, 4096, 16), (3, 200, 4096, 1100, 32)])
def test_gather_and_group_ops_match_oracle_across_kernel_forms(B, C, N, M, NS):
    """The channel-major gathers and their gradients over the shapes that select each kernel form: 16-byte index / output
    vectors or the ragged scalar form (npoint * nsample not a multiple of 4), channel counts that are not a multiple of the
    channel chunk, target rows in a 64 KiB / 128 KiB LDS table and rows too long for LDS (global-atomic fallback), hot
    indices (many gathers of one source point)."""
    from nsdp_amd import pointnet2_utils as pu
    feats = synth.normal(41, "feats", (B, C, N))
    idx = (synth.uniform01(42, "idx", (B, M)) * N).astype(np.int32)
    gidx = (synth.uniform01(43, "gidx", (B, M, NS)) * N).astype(np.int32)
    gidx[:, : M // 3, 0] = N - 1                   # a hot source point
    f = _dev(feats).requires_grad_(True)
    out = pu.gather_operation(f, _dev(idx))
    np.testing.assert_array_equal(out.detach().cpu().numpy(), ref.gather_points(feats, idx))
    go = synth.normal(44, "go", (B, C, M))
    out.backward(_dev(go))
    np.testing.assert_allclose(f.grad.cpu().numpy(), ref.gather_points_grad(go, idx, N), rtol=1e-5, atol=1e-5)
    f.grad = None
    out = pu.grouping_operation(f, _dev(gidx))
    np.testing.assert_array_equal(out.detach().cpu().numpy(), ref.group_points(feats, gidx))
    go = synth.normal(45, "go2", (B, C, M, NS))
    out.backward(_dev(go))
    want = ref.group_points_grad(go, gidx, N)
    np.testing.assert_allclose(f.grad.cpu().numpy(), want, rtol=1e-5, atol=1e-5 * max(1.0, float(np.abs(want).max())))


@pytest.mark.parametrize("B,c,m,n", [(2, 9, 1300, 700), (3, 16, 100, 501), (1, 3, 40000, 1000),
                                     # rows of n > 8192 targets: the sliced list form (4 / 2 channels per walk of the lists)
                                     (2, 6, 4096, 16384), (1, 5, 16384, 20000)])
def test_three_interpolate_kernel_forms(B, c, m, n):
    from nsdp_amd import pointnet2_utils as pu
    feats = synth.normal(51, "f", (B, c, m))
    idx = (synth.uniform01(52, "i", (B, n, 3)) * m).astype(np.int32)
    w = synth.uniform(53, "w", (B, n, 3), 0.0, 1.0)
    f = _dev(feats).requires_grad_(True)
    out = pu.three_interpolate(f, _dev(idx), _dev(w))
    np.testing.assert_array_equal(out.detach().cpu().numpy(), ref.three_interpolate(feats, idx, w))
    go = synth.normal(54, "go", (B, c, n))
    out.backward(_dev(go))
    np.testing.assert_allclose(f.grad.cpu().numpy(), ref.three_interpolate_grad(go, idx, w, m), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("radius,nsample", [(0.2, 8), (0.05, 4), (2.0, 16), (1e-4, 3)])
def test_ball_query_matches_oracle(radius, nsample):
    from nsdp_amd import pointnet2_utils as pu
    xyz = _cloud(21, 2, 1500)
    new_xyz = _cloud(22, 2, 300)
    got = pu.ball_query(radius, nsample, _dev(xyz), _dev(new_xyz))
    np.testing.assert_array_equal(got.cpu().numpy(), ref.ball_query(new_xyz, xyz, radius, nsample))


def _trace_names(fn):
    import ctypes
    from nsdp_amd import _lib
    L = _lib.lib()
    L.nsdp_trace_enable(1)
    try:
        out = fn()
    finally:
        L.nsdp_trace_enable(0)
    n = L.nsdp_trace_read(None, 0)
    buf = ctypes.create_string_buffer(n)
    L.nsdp_trace_read(buf, n)
    return out, set(buf.value.decode().split("\n"))


@pytest.mark.parametrize("kind", ["uniform", "dupes", "grid"])
@pytest.mark.parametrize("B,N,M,radius,nsample", [
    (2, 2048, 500, 0.2, 16), (3, 1500, 300, 0.05, 4), (2, 1000, 64, 2.0, 64), (2, 1025, 65, 0.3, 32), (1, 15, 7, 0.4, 8),
    (2, 16, 130, 1e-4, 3), (1, 3000, 1, 0.1, 1), (2, 4096, 200, 0.02, 32), (2, 700, 100, 0.25, 65)])
def test_ball_query_four_lane_scan_matches_oracle(kind, B, N, M, radius, nsample):
    """The four-lanes-per-query plane-tile scan (ball_query_quad_kernel) against the literal emulation of the reference kernel
    (ball_query_gpu.cu:9-44): first `nsample` hits in index order, the first hit in every unused slot, zeros without a hit --
    ragged sizes (tile / group / workgroup edges), early exits, exact distance ties, nsample above the LDS rows (fallback)."""
    from nsdp_amd import pointnet2_utils as pu
    xyz = _cloud(100 + N, B, N, kind)
    new_xyz = _cloud(200 + M, B, M) if kind != "grid" else np.ascontiguousarray(xyz[:, :M] + (0.0 if M <= N else 0.0))
    if new_xyz.shape[1] != M:
        new_xyz = _cloud(200 + M, B, M)
    got, names = _trace_names(lambda: pu.ball_query(radius, nsample, _dev(xyz), _dev(new_xyz)))
    np.testing.assert_array_equal(got.cpu().numpy(), ref.ball_query(new_xyz, xyz, radius, nsample))
    if os.environ.get("NSDP_SEARCH_QUAD", "1") != "0":
        assert ("ball_query_quad" in names) == (nsample <= 64), names


@pytest.mark.parametrize("kind", ["uniform", "dupes", "grid"])
@pytest.mark.parametrize("B,n,m", [(2, 500, 2048), (3, 700, 1300), (2, 65, 1025), (1, 7, 15), (2, 130, 16), (2, 100, 2),
                                   (1, 50, 1), (2, 2048, 4096), (1, 1, 3000)])
def test_three_nn_four_lane_scan_matches_oracle(kind, B, n, m):
    """three_nn_quad_kernel against the literal emulation of interpolate_gpu.cu:9-59 (bests in double from 1e40, strict `<` in
    index order): indices AND squared distances bit for bit, with exact ties and with fewer than three known points."""
    from nsdp_amd import _lib
    unknown = _cloud(300 + n, B, n)
    known = _cloud(400 + m, B, m, kind)
    if kind == "grid" and n <= m:
        unknown = np.ascontiguousarray(known[:, :n])          # queries ON lattice points: ties at every distance
    d2 = torch.empty(B, n, 3, device="cuda")
    idx = torch.empty(B, n, 3, dtype=torch.int32, device="cuda")
    u, k = _dev(unknown), _dev(known)

    def run():
        _lib.check(_lib.lib().nsdp_three_nn(_lib.fptr(u), _lib.fptr(k), B, n, m, _lib.fptr(d2), _lib.iptr(idx), _lib.stream_ptr()),
                   "nsdp_three_nn")
    _, names = _trace_names(run)
    rd2, ridx = ref.three_nn(unknown, known)
    np.testing.assert_array_equal(idx.cpu().numpy(), ridx)
    np.testing.assert_array_equal(d2.cpu().numpy(), rd2)
    if os.environ.get("NSDP_SEARCH_QUAD", "1") != "0":
        assert "three_nn_quad" in names, names


@pytest.mark.parametrize("B,n,m,k,sign", [(2, 300, 1000, 16, 1.0), (3, 100, 100, 100, 1.0), (2, 64, 500, 7, -1.0), (1, 1, 5, 3, -1.0)])
def test_rel_coords4_equals_gather_and_subtract(B, n, m, k, sign):
    """nsdp_rel_coords4 against the reference's index_points + broadcast subtraction (model/encoder/blocks.py:104-106,
    :285-286, model/decoder/blocks.py:72-78), bit for bit, with the fourth column zero."""
    from nsdp_amd import pointnet2_utils as pu
    q, s_ = _cloud(61 + n, B, n), _cloud(62 + m, B, m)
    idx = (synth.uniform01(63, "i", (B, n, k)) * m).astype(np.int32)
    got = pu.rel_coords4(_dev(q), _dev(s_), _dev(idx), sign).cpu().numpy()
    gathered = np.take_along_axis(s_[:, None, :, :].repeat(n, 1), idx[..., None].astype(np.int64), axis=2)
    want = q[:, :, None, :] - gathered
    if sign < 0:
        want = gathered - q[:, :, None, :]
    np.testing.assert_array_equal(got[..., :3], want)
    assert not got[..., 3].any()


def test_three_nn_and_interpolate_match_oracle():
    from nsdp_amd import pointnet2_utils as pu
    unknown, known = _cloud(31, 2, 700), _cloud(32, 2, 1300)
    dist, idx = pu.three_nn(_dev(unknown), _dev(known))
    rd2, ridx = ref.three_nn(unknown, known)
    np.testing.assert_array_equal(idx.cpu().numpy(), ridx)
    np.testing.assert_array_equal(dist.cpu().numpy(), np.sqrt(rd2))
    feats = synth.normal(33, "f", (2, 9, 1300))
    w = synth.uniform(34, "w", (2, 700, 3), 0.0, 1.0)
    f = _dev(feats).requires_grad_(True)
    out = pu.three_interpolate(f, idx, _dev(w))
    np.testing.assert_array_equal(out.detach().cpu().numpy(), ref.three_interpolate(feats, ridx, w))
    go = synth.normal(35, "go", (2, 9, 700))
    out.backward(_dev(go))
    np.testing.assert_allclose(f.grad.cpu().numpy(), ref.three_interpolate_grad(go, ridx, w, 1300),
                               rtol=1e-5, atol=1e-6)


def test_query_and_group_module():
    from nsdp_amd import pointnet2_utils as pu
    xyz, new_xyz = _cloud(41, 2, 400), _cloud(42, 2, 50)
    feats = synth.normal(43, "f", (2, 6, 400))
    out = pu.QueryAndGroup(0.3, 8)(_dev(xyz), _dev(new_xyz), _dev(feats))
    idx = ref.ball_query(new_xyz, xyz, 0.3, 8)
    gx = ref.group_points(np.ascontiguousarray(xyz.transpose(0, 2, 1)), idx) - new_xyz.transpose(0, 2, 1)[..., None]
    want = np.concatenate([gx, ref.group_points(feats, idx)], axis=1)
    np.testing.assert_array_equal(out.cpu().numpy(), want)


def test_rows_gather_scatter():
    from nsdp_amd import pointnet2_utils as pu
    for C in (3, 120, 256):
        pts = synth.normal(50 + C, "p", (2, 300, C))
        idx = (synth.uniform01(51 + C, "i", (2, 90)) * 300).astype(np.int32)
        out = pu.gather_rows(_dev(pts), _dev(idx)).cpu().numpy()
        np.testing.assert_array_equal(out, np.take_along_axis(pts, idx[..., None].astype(np.int64), axis=1))
        go = synth.normal(52 + C, "g", (2, 90, C))
        want = np.zeros_like(pts)
        for b in range(2):
            np.add.at(want[b], idx[b], go[b])
        got = pu.scatter_add_rows(_dev(go), _dev(idx), 300).cpu().numpy()
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-6)


def test_scatter_add_rows_inverse_list_form_matches_the_atomic_form():
    """index_points' backward at a size that takes the inverse-list + segment-sum path (deterministic) against numpy and
    against the atomic kernel (NSDP_SCATTER_ROWS=atomic's form)."""
    from nsdp_amd import pointnet2_utils as pu
    B, N, C, S = 3, 700, 128, 9000
    go = synth.normal(61, "g", (B, S, C))
    idx = (synth.uniform01(62, "i", (B, S)) * N).astype(np.int32)
    idx[:, :500] = 5                                    # a hot row; row 6 stays empty
    idx[idx == 6] = 7
    want = np.zeros((B, N, C), dtype=np.float64)
    for b in range(B):
        np.add.at(want[b], idx[b], go[b].astype(np.float64))
    a = pu.scatter_add_rows(_dev(go), _dev(idx), N)
    if pu._SCATTER_INVERSE:            # (NSDP_SCATTER_ROWS=atomic: the atomic kernel's order of summation varies)
        assert torch.equal(a, pu.scatter_add_rows(_dev(go), _dev(idx), N))          # deterministic
    was = pu._SCATTER_INVERSE
    pu._SCATTER_INVERSE = False
    try:
        b_ = pu.scatter_add_rows(_dev(go), _dev(idx), N)
    finally:
        pu._SCATTER_INVERSE = was
    tol = 1e-5 * float(np.abs(want).max())
    np.testing.assert_allclose(a.cpu().numpy(), want, rtol=1e-5, atol=tol)
    np.testing.assert_allclose(b_.cpu().numpy(), want, rtol=1e-5, atol=tol)
    assert float(a[:, 6].abs().max()) == 0.0
