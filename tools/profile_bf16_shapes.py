"""Every (shape, flags) the bf16-storage train step sends to the bf16 dense kernels, timed in isolation (events + sync
around each call), aggregated per shape.   python tools/profile_bf16_shapes.py [forward|arbitrary] [batch]"""
import collections, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from nsdp_amd import hip_linear, hip_linear_bf16 as hb, precision, synth
from nsdp_amd.model import build_model, optimizer_factory
from nsdp_amd.model.utils import compute_l2_error
workload = sys.argv[1] if len(sys.argv) > 1 else "forward"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
precision.set_storage("bf16")
hip_linear._OVERLAP_WGRAD = False
dev = torch.device("cuda:0")
cfg = bench.model_config()
if workload == "arbitrary":
    cfg["model"]["type"] = "arbitrary"
model, *_ = build_model(cfg, device="cpu")
state = synth.procedural_state_dict(model.state_dict(), 2048)
model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
model.to(dev).train()
_, opt = optimizer_factory({"optimizer": "Adam", "lr": 5e-4, "lr_step": 200, "lr_decay": 0.1, "weight_decay": 0.0}, model.parameters())
data = {k: torch.from_numpy(v).to(dev) for k, v in synth.make_batch(1000, B, bench.N_SURF, bench.N_QUERY).items()}
def forward():
    if workload == "arbitrary":
        s = data["surface_samples_inputs"]
        return model(data["space_samples_src"], s[:, :, 0:3], s[:, :, 3:6], s[:, :, 6:7])
    return model(data["space_samples_src"], data["surface_samples_inputs"])
def step():
    opt.zero_grad(set_to_none=True)
    loss = compute_l2_error(forward(), data["space_samples_tgt"])
    loss.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
def timed(key, nbytes, fn, *a, **k):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = fn(*a, **k); e1.record(); torch.cuda.synchronize()
    v = agg[key]; v[0] += 1; v[1] += e0.elapsed_time(e1) * 1e3; v[2] = nbytes
    return r
_run, _wg, _k4f, _k4w = hb.run, hb.wgrad, hb.k4_forward, hb.k4_wgrad
def run(x2, pack, N, b, residual, mask, out_mask, relu_in, relu_out, out_f32=False):
    M, K = x2.shape
    flags = ("r" if residual is not None else "") + ("m" if mask is not None else "") + ("o" if out_mask is not None else "") + ("f" if out_f32 else "")
    nb = 2 * M * (K + N) + (2 * M * N if residual is not None else 0) + (2 * M * K if mask is not None else 0) + (2 * M * N if out_mask is not None else 0)
    return timed(("linear", M, K, N, flags), nb, _run, x2, pack, N, b, residual, mask, out_mask, relu_in, relu_out, out_f32)
def wgrad(dy2, x2, mask, relu_x, want_db, out=None):
    M, N = dy2.shape; K = x2.shape[1]
    nb = 2 * M * (K + N) + (2 * M * N if mask is not None else 0)
    return timed(("wgrad", M, K, N, "m" if mask is not None else ""), nb, _wg, dy2, x2, mask, relu_x, want_db, out)
def k4f(x2, w4, b, relu_out):
    M = x2.shape[0]; N = w4.shape[0]
    return timed(("k4 fwd", M, 4, N, ""), M * (16 + 2 * N), _k4f, x2, w4, b, relu_out)
def k4w(dy2, x2, mask, relu_x, want_db):
    M, N = dy2.shape
    return timed(("k4 wgrad", M, 4, N, "m" if mask is not None else ""), M * (16 + 2 * N), _k4w, dy2, x2, mask, relu_x, want_db)
hb.run, hb.wgrad, hb.k4_forward, hb.k4_wgrad = run, wgrad, k4f, k4w
step()
torch.cuda.synchronize()
tot = sum(v[1] for v in agg.values())
print(f"{workload} bf16 B={B}: {sum(v[0] for v in agg.values())} bf16 dense launches per step, {tot/1e3:.2f} ms in isolation (wgrad includes its reduce)")
print(f"{'kind':9s} {'M':>8s} {'K':>4s} {'N':>4s} {'flags':5s} {'calls':>5s} {'us/call':>8s} {'ms/step':>8s} {'GB/s':>7s}")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k[0]:9s} {k[1]:8d} {k[2]:4d} {k[3]:4d} {k[4]:5s} {v[0]:5d} {v[1]/v[0]:8.1f} {v[1]/1e3:8.2f} {v[2]/(v[1]/v[0])/1e3:7.0f}")
