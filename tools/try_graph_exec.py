"""The captured train step replayed three ways: eager, torch's hipGraph replay, and this repo's multi-stream executor
(nsdp_amd/graph_step.py).  Losses must agree step for step.    python tools/try_graph_exec.py [batch] [dtype] [workload]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from nsdp_amd import precision, synth
from nsdp_amd.graph_step import GraphedStep, capturable_adam
from nsdp_amd.model import build_model, optimizer_factory
from nsdp_amd.model.utils import compute_l2_error
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
precision.set_storage(sys.argv[2] if len(sys.argv) > 2 else "f32")
workload = sys.argv[3] if len(sys.argv) > 3 else "forward"
dev = torch.device("cuda:0")
cfg = bench.model_config()
if workload == "arbitrary":
    cfg["model"]["type"] = "arbitrary"


def fresh(capturable):
    model, *_ = build_model(cfg, device="cpu")
    state = synth.procedural_state_dict(model.state_dict(), 2048)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
    model.to(dev).train()
    _, opt = optimizer_factory({"optimizer": "Adam", "lr": 5e-4, "lr_step": 200, "lr_decay": 0.1, "weight_decay": 0.0}, model.parameters())
    if capturable:
        capturable_adam(opt)
    return model, opt


data = {k: torch.from_numpy(v).to(dev) for k, v in synth.make_batch(1000, B, bench.N_SURF, bench.N_QUERY).items()}


def make_step(model, opt):
    def step():
        opt.zero_grad(set_to_none=True)
        s = data["surface_samples_inputs"]
        if workload == "arbitrary":
            pred = model(data["space_samples_src"], s[:, :, 0:3], s[:, :, 3:6], s[:, :, 6:7])
        else:
            pred = model(data["space_samples_src"], s)
        loss = compute_l2_error(pred, data["space_samples_tgt"])
        loss.backward()
        opt.step()
        return loss
    return step


def clock(fn, n=10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n, 1e3 * t_enq / n


model, opt = fresh(False)
step = make_step(model, opt)
eager = [float(step().item()) for _ in range(8)]
t_eager = clock(step)
torch.cuda.synchronize()
t0 = time.perf_counter(); step(); t_host_e = 1e3 * (time.perf_counter() - t0)
torch.cuda.synchronize()
print(f"host time of one eager step on an idle GPU: {t_host_e:.2f} ms")
print("eager   ", [f"{v:.6f}" for v in eager], f"{t_eager[0]:.2f} ms/step (host enqueue {t_eager[1]:.2f})")
del model, opt, step

model, opt = fresh(True)
gs = GraphedStep(make_step(model, opt), max_streams=int(os.environ.get("NSDP_GRAPH_STREAMS", "4")))
gs.capture(warmup=3)
print("executor:", gs.info)
losses = []
# capture() ran 3 eager steps + the captured one does NOT execute: the next replay is step 4
for _ in range(5):
    losses.append(float(gs().item()))
print("replayed", ["--------"] * 3 + [f"{v:.6f}" for v in losses])
t_exec = clock(gs)
torch.cuda.synchronize()
t0 = time.perf_counter(); gs(); t_host = 1e3 * (time.perf_counter() - t0)
torch.cuda.synchronize()
print(f"host time of one replay call on an idle GPU: {t_host:.2f} ms")
print(f"multi-stream executor {t_exec[0]:.2f} ms/step (host enqueue {t_exec[1]:.2f})")
torch.cuda.synchronize()
gs._graph.instantiate()
t_torch = clock(gs._graph.replay)
print(f"hipGraphLaunch (torch replay) {t_torch[0]:.2f} ms/step (host {t_torch[1]:.2f});  eager {t_eager[0]:.2f}")
ok = all(abs(a - b) <= 2e-3 * abs(a) + 1e-6 for a, b in zip(eager[3:8], losses))
print("losses of steps 4..8 agree with eager:", ok)
