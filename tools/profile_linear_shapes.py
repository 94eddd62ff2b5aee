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
hip_linear._run, hip_linear._wgrad = rec_fwd, rec_wgrad
for _ in range(2):
    calls.clear()
    model.zero_grad(set_to_none=True)
    loss = compute_l2_error(model(data["space_samples_src"], data["surface_samples_inputs"]), data["space_samples_tgt"])
    loss.backward()
torch.cuda.synchronize()
hip_linear._run, hip_linear._wgrad, hip_linear._k4tail_fn = orig_fwd, orig_wgrad, orig_tail

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
    rows.append((t * cnt, key[0], M, N, K, flags, cnt, t, fl / t / 1e9))
    del x
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"total isolated GEMM time per step: {tot:.2f} ms  (nt {sum(r[0] for r in rows if r[1].startswith('nt')):.2f}, k4 tail {sum(r[0] for r in rows if r[1]=='tail-x3'):.2f}, wgrad {sum(r[0] for r in rows if r[1]=='wgrad'):.2f})")
print("kind   M        N    K    flags  count  ms/call  TF     ms/step  cum%")
cum = 0.0
for tt, kind, M, N, K, flags, cnt, t, tf in rows:
    cum += tt
    print(f"{kind:6s} {M:8d} {N:4d} {K:4d} {flags:6s} {cnt:5d}  {t:7.3f}  {tf:6.1f} {tt:7.2f}  {100*cum/tot:5.1f}")
