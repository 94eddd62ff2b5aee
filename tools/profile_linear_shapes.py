"""Which dense-layer shapes carry the train step?  Records every (kind, M, N, K, flags) the B=32 forward.yaml
train step sends to the linear / wgrad kernels, then times each distinct configuration in isolation.
    python tools/profile_linear_shapes.py [--batch 32]
"""
import argparse, collections, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from nsdp_amd import hip_linear, synth
from nsdp_amd.model import build_model, optimizer_factory
from nsdp_amd.model.utils import compute_l2_error

ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=32); args = ap.parse_args()
dev = torch.device("cuda:0")
cfg = bench.model_config()
model, *_ = build_model(cfg, device="cpu")
state = synth.procedural_state_dict(model.state_dict(), 2048)
model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
model.to(dev).train()
data = {k: torch.from_numpy(v).to(dev) for k, v in synth.make_batch(1000, args.batch, bench.N_SURF, bench.N_QUERY).items()}

calls = collections.Counter()
orig_fwd, orig_wgrad = hip_linear._run, hip_linear._wgrad
def rec_fwd(kind, x2, wp, N, b, residual, mask, out_mask, relu_in, relu_out, *a, **kw):
    calls[("nt-" + kind, x2.shape[0], N, x2.shape[1], b is not None, residual is not None, mask is not None, out_mask is not None, bool(relu_in), bool(relu_out))] += 1
    return orig_fwd(kind, x2, wp, N, b, residual, mask, out_mask, relu_in, relu_out, *a, **kw)
orig_tail = hip_linear._k4tail_fn
def rec_tail(link, wpt, n_hidden, kind_t, h0):
    fn = orig_tail(link, wpt, n_hidden, kind_t, h0)
    def wrapped(dy2, x4, mask, relu_x, want_db, out=None):
        M, K = dy2.shape
        if hip_linear.K4_TAIL and kind_t == "x3" and hip_linear.lib().nsdp_linear_bf16x3_k4tail_ok(hip_linear._ll(M), hip_linear._ci(n_hidden), hip_linear._ci(K)):
            calls[("tail-x3", M, n_hidden, K)] += 1      # the dX GEMM with the K = 4 weight gradient in its epilogue: no output
        return fn(dy2, x4, mask, relu_x, want_db, out)
    return wrapped
hip_linear._k4tail_fn = rec_tail
def rec_wgrad(dy2, x2, mask, relu_x, want_db, out=None):
    calls[("wgrad", dy2.shape[0], dy2.shape[1], x2.shape[1], mask is not None, bool(relu_x), bool(want_db))] += 1
    return orig_wgrad(dy2, x2, mask, relu_x, want_db, out)
orig_gather = hip_linear._fwd_x3_gather
def rec_gather(x2, wp, N, b, gather, relu_in, relu_out):
    # the position-encoding GEMM whose epilogue adds the gathered q - k rows (nsdp_linear_bf16x3_gather_f32): the same kernel class
    calls[("gat-x3", x2.shape[0], N, x2.shape[1], b is not None, gather[0] is not None, bool(relu_in), bool(relu_out),
           int(gather[1]), int(gather[4]), int(gather[5]))] += 1
    return orig_gather(x2, wp, N, b, gather, relu_in, relu_out)
hip_linear._run, hip_linear._wgrad, hip_linear._fwd_x3_gather = rec_fwd, rec_wgrad, rec_gather
for _ in range(2):
    calls.clear()
    model.zero_grad(set_to_none=True)
    loss = compute_l2_error(model(data["space_samples_src"], data["surface_samples_inputs"]), data["space_samples_tgt"])
    loss.backward()
torch.cuda.synchronize()
hip_linear._run, hip_linear._wgrad, hip_linear._k4tail_fn, hip_linear._fwd_x3_gather = orig_fwd, orig_wgrad, orig_tail, orig_gather

def timeit(fn, n=8):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

rows = []
for key, cnt in calls.items():
    if key[0] == "tail-x3":
        _, M, N, K = key
        x = torch.relu(torch.randn(M, K, device=dev))
        lin0 = torch.nn.Linear(3, N).to(dev)
        wpt = hip_linear.pack_weight_x3(torch.randn(K, N, device=dev), True, True)[1]
        x4 = torch.nn.functional.pad(torch.randn(M, 3, device=dev), (0, 1)).contiguous()
        link = hip_linear.K4Tail(); link.w_param, link.b_param, link.k_orig = lin0.weight, lin0.bias, 3
        tfn = orig_tail(link, wpt, N, "x3", None)
        t = timeit(lambda: tfn(x, x4, None, False, True))
        flags = "k4"
    elif key[0] == "gat-x3":
        _, M, N, K, hb, two, ri, ro, kk, rps, nsrc = key
        x = torch.randn(M, K, device=dev)
        wp = hip_linear.pack_weight_x3(torch.randn(N, K, device=dev))[0]
        b = torch.randn(N, device=dev) if hb else None
        # the step's own geometry: rows = (shape, centre, neighbour); a per-centre q table + a per-shape k table (two tables), or
        # one per-shape table of differences (the decoder: one query vector per shape)
        gidx = torch.randint(0, nsrc, (M,), device=dev, dtype=torch.int32)
        gk = torch.randn((M // rps) * nsrc, N, device=dev)
        gq = torch.randn(-(-M // kk), N, device=dev) if two else None
        t = timeit(lambda: orig_gather(x, wp, N, b, (gq, kk, gk, gidx, rps, nsrc), ri, ro))
        flags = "".join(c for c, f in zip("b2IO", (hb, two, ri, ro)) if f)
    elif key[0].startswith("nt"):
        kname, M, N, K, hb, hr, hm, ho, ri, ro = key
        kind = kname[3:]
        x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev)
        wp = (hip_linear.pack_weight_x3 if kind == "x3" else hip_linear.pack_weight)(w)[0]
        b = torch.randn(N, device=dev) if hb else None
        r = torch.randn(M, N, device=dev) if hr else None
        m = torch.randn(M, K, device=dev) if hm else None
        o = torch.randn(M, N, device=dev) if ho else None
        t = timeit(lambda: orig_fwd(kind, x, wp, N, b, r, m, o, ri, ro))
        flags = "".join(c for c, f in zip("brmoIO", (hb, hr, hm, ho, ri, ro)) if f)
    else:
        _, M, N, K, hm, rx, wdb = key
        dy = torch.randn(M, N, device=dev); x = torch.randn(M, K, device=dev)
        m = torch.randn(M, N, device=dev) if hm else None
        t = timeit(lambda: orig_wgrad(dy, x, m, rx, wdb))
        flags = "".join(c for c, f in zip("mxb", (hm, rx, wdb)) if f)
    fl = 2.0 * M * N * K
    # algorithmic bytes per launch exactly as the C side accounts them (4(M(K+N)+NK); the tail form stores nothing:
    # 4(M(K+4)+NK); weight gradient 4M(N+K)) -- csrc/gemm_bf16x3.hip, csrc/wgrad_bf16x3.hip prof::Scope lines
    ab = 4.0 * (M * (K + 4) + N * K) if key[0] == "tail-x3" else 4.0 * M * (N + K) if key[0] == "wgrad" else 4.0 * (M * (K + N) + N * K)
    rows.append((t * cnt, key[0], M, N, K, flags, cnt, t, fl / t / 1e9, ab))
    del x
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"total isolated GEMM time per step: {tot:.2f} ms  (nt {sum(r[0] for r in rows if r[1].startswith('nt')):.2f}, gather epilogue {sum(r[0] for r in rows if r[1]=='gat-x3'):.2f}, k4 tail {sum(r[0] for r in rows if r[1]=='tail-x3'):.2f}, wgrad {sum(r[0] for r in rows if r[1]=='wgrad'):.2f})")
# per kernel class: launches, algorithmic bytes, isolated time per step -- `linear_bf16x3_kernel` here must equal the bench line's
# roofline.launches / 2 (the isolated pass times two steps) and roofline.algorithmic_bytes x launches
classes = {"linear_bf16x3_kernel": ("nt-x3", "gat-x3", "tail-x3"), "linear_nt_kernel (exact fp32)": ("nt-wp",), "weight gradients (all kernels)": ("wgrad",)}
print("class                              launches/step  algorithmic GB/step  isolated ms/step  GB/s    frac of 8 TB/s")
for cname, kinds in classes.items():
    sel = [r for r in rows if r[1] in kinds]
    n = sum(r[6] for r in sel); gb = sum(r[9] * r[6] for r in sel) / 1e9; ms = sum(r[0] for r in sel)
    if n:
        print(f"{cname:34s} {n:13d}  {gb:19.2f}  {ms:16.2f}  {gb / ms * 1e3:6.0f}  {gb / ms / 8.0:6.3f}")
print("kind   M        N    K    flags  count  ms/call  TF     ms/step  cum%   alg MB/launch")
cum = 0.0
for tt, kind, M, N, K, flags, cnt, t, tf, ab in rows:
    cum += tt
    print(f"{kind:6s} {M:8d} {N:4d} {K:4d} {flags:6s} {cnt:5d}  {t:7.3f}  {tf:6.1f} {tt:7.2f}  {100*cum/tot:5.1f}  {ab / 1e6:9.2f}")
