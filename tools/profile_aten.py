"""Which ATen kernels remain in the train step, on what shapes and from which call site (torch profiler, one step).
    python tools/profile_aten.py [forward|arbitrary] [f32|bf16] [batch]"""
import os, sys, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from nsdp_amd import precision, synth
from nsdp_amd.model import build_model, optimizer_factory
from nsdp_amd.model.utils import compute_l2_error
workload = sys.argv[1] if len(sys.argv) > 1 else "forward"
precision.set_storage(sys.argv[2] if len(sys.argv) > 2 else "f32")
B = int(sys.argv[3]) if len(sys.argv) > 3 else 32
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
for _ in range(2): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step(); torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
def site(e):
    for fr in e.stack or []:
        if "/nsdp_amd/" in fr and "torch/" not in fr:
            return fr.split("/nsdp_amd/")[1][:60]
    return "(autograd engine / optimizer)"
for e in prof.events():
    if e.device_time > 0 and e.name.startswith("aten::"):
        k = (e.name, str(e.input_shapes)[:70], site(e))
        agg[k][0] += 1; agg[k][1] += e.device_time
tot = sum(v[1] for v in agg.values()); n = sum(v[0] for v in agg.values())
print(f"{workload} {precision.storage_dtype()} B={B}: aten device time per step {tot / 1e3:.2f} ms in {n} ops")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:45]:
    print(f"{v[1]/1e3:7.3f} ms {v[0]:4d}x {k[0]:24s} {k[1]:70s} {k[2]}")
