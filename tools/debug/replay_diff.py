"""Which gradients of a replayed step differ from the eager step's?  python tools/debug/replay_diff.py arbitrary f32 8 full"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import build_product, model_cfg, restore_model, snapshot_model, to_dev
from nsdp_amd import precision, synth
from nsdp_amd.graph_step import GraphedStep, capturable_adam
from nsdp_amd.model import optimizer_factory
mtype, dtype, B, size = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
npl, ns, nq = ([2048, 500, 100], 2048, 8192) if size == "full" else ([256, 64, 16], 256, 128)
DEV = torch.device("cuda:0")
precision.set_storage(dtype)
cfg = model_cfg(mtype, npl)
data = to_dev(synth.make_batch(193, B, ns, nq), DEV)
model, train_fn, _ = build_product(cfg, 193, DEV)
model.train()
_, opt = optimizer_factory({"optimizer": "Adam", "lr": 5e-5}, model.parameters())
step = lambda: train_fn.tensor_step(model, opt, data, cfg)
capturable_adam(opt)
snap = snapshot_model(model)
step(); torch.cuda.synchronize()
def one(run):
    restore_model(model, snap, opt)
    loss = run(); torch.cuda.synchronize()
    return loss.detach().clone(), {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
l0, g0 = one(step)
l1, g1 = one(step)
print("eager twice equal:", torch.equal(l0, l1), sum(not torch.equal(g0[k], g1[k]) for k in g0), "grads differ")
restore_model(model, snap, opt)
gs = GraphedStep(step, max_streams=int(os.environ.get("STREAMS", "1"))).capture(warmup=0)
l2, g2 = one(gs)
bad = [k for k in g0 if not torch.equal(g0[k], g2[k])]
print("replay vs eager: loss equal", torch.equal(l0, l2), len(bad), "of", len(g0), "grads differ")
names = [k for k, _ in model.named_parameters() if k in g0]
print("EQUAL:", [k for k in names if k not in bad][:60])
for k in reversed(names):
    if k in bad:
        d = (g0[k] - g2[k]).abs().max().item(); print("last differing (first in backward order?):", k, d, g0[k].abs().max().item()); break
worst = sorted(bad, key=lambda k: -((g0[k] - g2[k]).abs().max() / (g0[k].abs().max() + 1e-30)).item())[:5]
for k in worst:
    print("worst rel:", k, ((g0[k] - g2[k]).abs().max() / (g0[k].abs().max() + 1e-30)).item())
