"""tests/golden/pointnet2_modules.npz: the reference's pointnet2_ops/pointnet2_modules.py run on CPU, with its CUDA
extension `pointnet2_ops._ext` replaced by the literal kernel emulation of oracle/pointnet2_ref.c (the reference's
Python -- autograd Functions, QueryAndGroup, SA / FP modules -- is imported unmodified).
Run in the build container only:  python oracle/make_golden_modules.py"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nsdp_amd import synth  # noqa: E402
from oracle import pointnet2_ref as R  # noqa: E402


def _stub_ext():
    t = torch.from_numpy
    n = lambda x: x.detach().cpu().numpy()
    ext = types.ModuleType("pointnet2_ops._ext")
    ext.furthest_point_sampling = lambda p, k: t(R.furthest_point_sampling(n(p), int(k)))
    ext.gather_points = lambda p, i: t(R.gather_points(n(p), n(i)))
    ext.gather_points_grad = lambda g, i, m: t(R.gather_points_grad(n(g), n(i), int(m)))
    ext.group_points = lambda p, i: t(R.group_points(n(p), n(i)))
    ext.group_points_grad = lambda g, i, m: t(R.group_points_grad(n(g), n(i), int(m)))
    ext.ball_query = lambda nx, x, r, k: t(R.ball_query(n(nx), n(x), float(r), int(k)))
    ext.three_nn = lambda u, k: tuple(t(a) for a in R.three_nn(n(u), n(k)))
    ext.three_interpolate = lambda p, i, w: t(R.three_interpolate(n(p), n(i), n(w)))
    ext.three_interpolate_grad = lambda g, i, w, m: t(R.three_interpolate_grad(n(g), n(i), n(w), int(m)))
    sys.modules["pointnet2_ops._ext"] = ext


def main():
    R.build()
    _stub_ext()
    sys.path.insert(0, "/root/reference/pointnet2_ops_lib")
    from pointnet2_ops import pointnet2_modules as M
    torch.manual_seed(0)
    B, N, C = 2, 256, 16
    xyz = torch.from_numpy(synth.uniform(9, "xyz", (B, N, 3), -0.5, 0.5))
    feats = torch.from_numpy(synth.normal(9, "feats", (B, C, N)))
    fx = {"xyz": xyz.numpy(), "feats": feats.numpy()}

    seeds = {}
    for tag, ctor, fwd in [
        ("msg", lambda: M.PointnetSAModuleMSG(64, [0.15, 0.3], [8, 16], [[C, 32, 48], [C, 32, 64]]), lambda m, f: m(xyz, f)),
        ("sa_all", lambda: M.PointnetSAModule([C, 64, 96]), lambda m, f: m(xyz, f)),
        ("fp", lambda: M.PointnetFPModule([C + 8, 64, 32]), None),
    ]:
        mod = ctor()
        seed = 500 + len(seeds)
        seeds[tag] = seed
        state = synth.procedural_state_dict(mod.state_dict(), seed)
        mod.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
        mod.train()
        f = feats.clone().requires_grad_(True)
        if tag == "fp":
            unknown = torch.from_numpy(synth.uniform(9, "unk", (B, 400, 3), -0.5, 0.5))
            ufe = torch.from_numpy(synth.normal(9, "ufe", (B, 8, 400)))
            fx["fp/unknown"], fx["fp/ufe"] = unknown.numpy(), ufe.numpy()
            out = mod(unknown, xyz, ufe, f)
        else:
            new_xyz, out = mod(xyz, f)
            if new_xyz is not None:
                fx[tag + "/new_xyz"] = new_xyz.numpy()
        go = torch.from_numpy(synth.normal(9, tag + "go", tuple(out.shape)))
        out.backward(go)
        fx[tag + "/out"], fx[tag + "/go"], fx[tag + "/dfeats"] = out.detach().numpy(), go.numpy(), f.grad.numpy()
        fx[tag + "/seed"] = np.int64(seed)
        for k, p in mod.named_parameters():
            fx[tag + "/grad/" + k] = p.grad.numpy()
        for k, v in mod.state_dict().items():
            if k.endswith(("running_mean", "running_var")):
                fx[tag + "/bn/" + k] = v.numpy()
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "pointnet2_modules.npz"), **fx)
    print("wrote pointnet2_modules.npz", {k: v.shape for k, v in fx.items() if k.endswith("/out")})


if __name__ == "__main__":
    main()
