"""Whole-step hipGraph capture experiment: does a captured train step train?  python tools/try_graph.py [batch] [dtype]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from nsdp_amd import precision, synth
from nsdp_amd.model import build_model, optimizer_factory
from nsdp_amd.model.utils import compute_l2_error
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
precision.set_storage(sys.argv[2] if len(sys.argv) > 2 else "f32")
dev = torch.device("cuda:0")
cfg = bench.model_config()


def fresh():
    model, *_ = build_model(cfg, device="cpu")
    state = synth.procedural_state_dict(model.state_dict(), 2048)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
    model.to(dev).train()
    _, opt = optimizer_factory({"optimizer": "Adam", "lr": 5e-4, "lr_step": 200, "lr_decay": 0.1, "weight_decay": 0.0}, model.parameters())
    return model, opt


data = {k: torch.from_numpy(v).to(dev) for k, v in synth.make_batch(1000, B, bench.N_SURF, bench.N_QUERY).items()}


def make_step(model, opt):
    def step():
        opt.zero_grad(set_to_none=True)
        loss = compute_l2_error(model(data["space_samples_src"], data["surface_samples_inputs"]), data["space_samples_tgt"])
        loss.backward()
        opt.step()
        return loss
    return step


# eager reference: losses of steps 1..8
model, opt = fresh()
step = make_step(model, opt)
eager = [float(step().item()) for _ in range(8)]
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    step()
torch.cuda.synchronize()
t_eager = (time.perf_counter() - t0) / 10
print("eager losses", [f"{v:.6f}" for v in eager], f"{1e3 * t_eager:.2f} ms/step")

model, opt = fresh()
for g in opt.param_groups:
    g["capturable"] = True
step = make_step(model, opt)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
losses = []
with torch.cuda.stream(side):
    for _ in range(3):
        losses.append(step())
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
losses = [float(l.item()) for l in losses]
w0 = [p.detach().clone() for p in model.parameters()]
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    static_loss = step()
torch.cuda.synchronize()
w1 = [p.detach().clone() for p in model.parameters()]
print("changed during capture:", sum(int(not torch.equal(a, b)) for a, b in zip(w0, w1)), "of", len(w0))
for i in range(5):
    graph.replay()
    torch.cuda.synchronize()
    losses.append(float(static_loss.item()))
w2 = [p.detach().clone() for p in model.parameters()]
print("changed by 5 replays:", sum(int(not torch.equal(a, b)) for a, b in zip(w1, w2)), "of", len(w1))
print("graph losses", [f"{v:.6f}" for v in losses])
t0 = time.perf_counter()
for _ in range(10):
    graph.replay()
torch.cuda.synchronize()
print(f"graph replay {1e3 * (time.perf_counter() - t0) / 10:.2f} ms/step   eager {1e3 * t_eager:.2f}")
