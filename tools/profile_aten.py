"""Which ATen kernels remain in the train step and on what shapes (torch profiler, one step)."""
import os, sys, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from nsdp_amd import synth
from nsdp_amd.model import build_model, optimizer_factory
from nsdp_amd.model.utils import compute_l2_error
dev = torch.device("cuda:0")
cfg = bench.model_config()
model, *_ = build_model(cfg, device="cpu")
state = synth.procedural_state_dict(model.state_dict(), 2048)
model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
model.to(dev).train()
_, opt = optimizer_factory({"optimizer": "Adam", "lr": 5e-4, "lr_step": 200, "lr_decay": 0.1, "weight_decay": 0.0}, model.parameters())
data = {k: torch.from_numpy(v).to(dev) for k, v in synth.make_batch(1000, 32, bench.N_SURF, bench.N_QUERY).items()}
def step():
    opt.zero_grad(set_to_none=True)
    loss = compute_l2_error(model(data["space_samples_src"], data["surface_samples_inputs"]), data["space_samples_tgt"])
    loss.backward(); opt.step()
for _ in range(2): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if e.device_time > 0 and e.name.startswith("aten::"):
        k = (e.name, str(e.input_shapes)[:90])
        agg[k][0] += 1; agg[k][1] += e.device_time
tot = sum(v[1] for v in agg.values())
print("aten device time per step: %.2f ms" % (tot / 1e3))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
    print(f"{v[1]/1e3:7.3f} ms {v[0]:4d}x {k[0]:28s} {k[1]}")
