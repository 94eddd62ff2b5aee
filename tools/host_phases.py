"""Host time per phase of a train step (no syncs inside; the GPU runs behind): forward / backward / optimizer.
    python tools/host_phases.py [forward|arbitrary] [f32|bf16] [batch]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from nsdp_amd import precision, synth
from nsdp_amd.model import build_model, optimizer_factory
from nsdp_amd.model.utils import compute_l2_error
workload = sys.argv[1] if len(sys.argv) > 1 else "forward"
precision.set_storage(sys.argv[2] if len(sys.argv) > 2 else "f32")
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
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
acc = [0.0, 0.0, 0.0, 0.0]
def step(record):
    t0 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    loss = compute_l2_error(forward(), data["space_samples_tgt"])
    t1 = time.perf_counter()
    loss.backward()
    t2 = time.perf_counter()
    opt.step()
    t3 = time.perf_counter()
    if record:
        acc[0] += t1 - t0; acc[1] += t2 - t1; acc[2] += t3 - t2
for _ in range(5): step(False)
torch.cuda.synchronize()
import gc; gc.collect(); gc.disable()
n = 10
t0 = time.perf_counter()
for _ in range(n): step(True)
te = time.perf_counter() - t0
torch.cuda.synchronize()
tw = time.perf_counter() - t0
print(f"{workload} {sys.argv[2] if len(sys.argv) > 2 else 'f32'} B={B}: host forward {1e3*acc[0]/n:.2f} ms, backward {1e3*acc[1]/n:.2f} ms, optimizer {1e3*acc[2]/n:.2f} ms; "
      f"enqueue {1e3*te/n:.2f} ms/step, wall {1e3*tw/n:.2f} ms/step")
