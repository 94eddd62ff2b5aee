import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np, torch
from helpers import *
from test_model_gpu import _hooks
DEV = torch.device('cuda:0')
fx, cfg, seed, data = fixture_setup("full_forward", "forward")
model, _, _ = build_product(cfg, seed, DEV)
model.eval()
tape = {}
hs = _hooks(model, tape)
with torch.no_grad():
    out = run_forward(model, cfg, to_dev(data, DEV))
o = out.cpu().numpy(); r = fx["eval_out"]
e = np.sqrt(((o - r) ** 2).sum(-1))[0]
print("per-point err: max", e.max(), "median", np.median(e), "n>1e-4", (e > 1e-4).sum(), "of", e.size)
print("worst idx", np.argsort(-e)[:10], np.sort(-e)[:10])
for key, ref in fx.items():
    if key.startswith("eval_tap/"):
        mine = sample_flat(tape[key[len("eval_tap/"):]], 64)
        print(key, float(np.abs(mine - ref).max()), float(np.abs(ref).max()))
from nsdp_amd import pointnet2_utils as pu
xyz0 = to_dev(data, DEV)["surface_samples_inputs"][:, :, :3].contiguous()
fps1 = pu.furthest_point_sample(xyz0, 500); print("fps1 eq", np.array_equal(fps1.cpu().numpy(), fx["geo/fps1"]))
xyz1 = pu.gather_rows(xyz0, fps1); fps2 = pu.furthest_point_sample(xyz1, 100); print("fps2 eq", np.array_equal(fps2.cpu().numpy(), fx["geo/fps2"]))
xyz2 = pu.gather_rows(xyz1, fps2)
q = to_dev(data, DEV)["space_samples_src"]
sites = {"begin": (xyz0, xyz0, 10), "tsa0": (xyz1, xyz0, 16), "down0": (xyz1, xyz1, 16), "tsa1": (xyz2, xyz1, 16), "down1": (xyz2, xyz2, 16), "dec": (q, xyz2, 7)}
for name, (a, s, k) in sites.items():
    idx = pu.knn(a, s, k).cpu().numpy(); ref = fx["geo/knn_" + name]
    st = max(1, idx.shape[1] // 64)
    print(name, "knn eq", np.array_equal(idx[:, ::st], ref))
