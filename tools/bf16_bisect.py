#!/usr/bin/env python
"""Where does the eval-output error of bf16 STORAGE (BASELINE config 3) come from?  GPU box only.

Eval forward of forward.yaml (B = 1) and arbitrary.yaml (B = 2) at 2048 surface + 8192 query points with the procedural
weights and seeded inputs of tests/golden/full_forward.npz / full_arbitrary.npz, compared with (a) the REFERENCE's fp32
output in the fixture and (b) this repo's fp32 output:

  * chaos floor: the fp32 product on input coordinates rounded to bf16 (relative error <= 2^-9 with random sign: what bf16
    storage of the INPUTS alone would do -- neighbour sets flip, FPS picks other centres), and on inputs x (1 + 2^-e) with
    the output divided by the same factor (the uniform variant: the map is nearly scale-equivariant);
  * all-bf16 storage, and bf16 storage with one part of the network at a time held in fp32 storage (module forward run
    under precision.storage(f32), casts at its boundary): both networks' encoders, decoders' attention block, decoders'
    residual trunk; and for FlowArbitrary each of the two networks as a whole.

Metric: max over shapes of sqrt(mean_q |delta|^2) without the two worst queries per shape (the reference's unstable
argsort, as in bench.py / tests)."""
import contextlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from nsdp_amd import precision, synth  # noqa: E402
from nsdp_amd.model import build_model  # noqa: E402
from nsdp_amd.model.decoder import crosstransformer_decoder as ctd  # noqa: E402

DEV = torch.device("cuda:0")
F32, BF = torch.float32, torch.bfloat16


def l2(a, b):
    err = ((a.astype(np.float64) - b.astype(np.float64)) ** 2).sum(-1)
    return float(np.sqrt(np.sort(err, axis=1)[:, :-2].mean(-1)).max())


@contextlib.contextmanager
def held_f32(modules, cast_out=True):
    """Run the forward of every module in `modules` in fp32 storage (floating inputs cast up, outputs cast back down)."""
    saved = []

    def up(t):
        return t.float() if (torch.is_tensor(t) and t.is_floating_point()) else t

    def down(t, like_coords=False):
        if torch.is_tensor(t) and t.is_floating_point() and t.shape[-1] != 3 and cast_out:
            return t.to(precision.storage_dtype())
        return t
    for m in modules:
        orig = m.forward

        def fwd(*a, _orig=orig, **k):
            a = [({kk: up(vv) for kk, vv in x.items()} if isinstance(x, dict) else up(x)) for x in a]
            with precision.storage(F32):
                out = _orig(*a, **k)
            if isinstance(out, dict):
                return {kk: down(vv) for kk, vv in out.items()}
            if isinstance(out, tuple):
                return tuple(down(v) for v in out)
            return down(out)
        m.forward = fwd
        saved.append((m, orig))
    try:
        yield
    finally:
        for m, orig in saved:
            m.forward = orig


def nets(model, mtype):
    return [model.model_canonicalize, model.model_deform] if mtype == "arbitrary" else [model]


def main():
    for name, mtype in (("full_forward", "forward"), ("full_arbitrary", "arbitrary")):
        fx = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"), allow_pickle=False)
        seed, b, ns, nq = (int(fx[k]) for k in ("meta_seed", "meta_batch", "meta_ns", "meta_nq"))
        from helpers import model_cfg
        cfg = model_cfg(mtype, fx["meta_npl"])
        model, *_ = build_model(cfg, device="cpu")
        state = synth.procedural_state_dict(model.state_dict(), seed)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
        model.to(DEV).eval()
        d = {k: torch.from_numpy(v).to(DEV) for k, v in synth.make_batch(seed, b, ns, nq).items()}
        stride = int(fx["meta_eval_stride"]) if "meta_eval_stride" in fx else 1
        ref = fx["eval_out"]

        def run(scale=1.0, round_inputs=False):
            s_in = d["surface_samples_inputs"] * scale
            q = d["space_samples_src"] * scale
            if round_inputs:          # the coordinates as a bf16 tensor would hold them (relative error <= 2^-9, random sign)
                s_in, q = s_in.to(BF).float(), q.to(BF).float()
            with torch.no_grad():
                if mtype == "arbitrary":
                    o = model(q, s_in[:, :, 0:3].contiguous(), s_in[:, :, 3:6].contiguous(), s_in[:, :, 6:7].contiguous() / scale)
                else:
                    s2 = s_in.clone()
                    s2[:, :, 6:7] = d["surface_samples_inputs"][:, :, 6:7]
                    o = model(q, s2)
            return o.float().cpu().numpy()[:, ::stride]

        print(f"== {mtype}.yaml eval, B = {b}, {ns} surface + {nq} query points ==")
        with precision.storage(F32):
            base = run()
            print(f"  fp32 product vs reference                           : {l2(base, ref):.3e}")
            print(f"  chaos floor: fp32 on bf16-ROUNDED input coordinates : {l2(run(round_inputs=True), base):.3e}")
            for e in (8, 12, 16):
                pert = run(1.0 + 2.0 ** -e) / (1.0 + 2.0 ** -e)
                print(f"  chaos floor: fp32, inputs x (1 + 2^-{e:<2d}), output / it : {l2(pert, base):.3e}")
        ns_ = nets(model, mtype)
        cases = [("all bf16", [], False)]
        if mtype == "arbitrary":
            cases += [("network 1 (canonicalize) in fp32", [ns_[0]], False), ("network 2 (deform) in fp32", [ns_[1]], False)]
        cases += [("encoders in fp32", [n.encoder for n in ns_], False),
                  ("decoder attention blocks (ct1) in fp32", [n.decoder.ct1 for n in ns_], False),
                  ("decoder trunks in fp32", [], True),
                  ("decoders (attention + trunk) in fp32", [n.decoder for n in ns_], False),
                  ("encoders + trunks in fp32", [n.encoder for n in ns_], True)]
        with precision.storage(BF):
            for label, mods, trunk in cases:
                ctd.TRUNK_F32 = trunk
                with held_f32(mods):
                    o = run()
                ctd.TRUNK_F32 = False
                print(f"  bf16 storage, {label:<40s}: vs reference {l2(o, ref):.3e}   vs fp32 product {l2(o, base):.3e}")


if __name__ == "__main__":
    main()
