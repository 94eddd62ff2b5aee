"""Generates tests/golden/dataset_contract.npz with the REFERENCE's dataset/utils.py (imported from /root/reference,
`trimesh` stubbed) on procedural samples: sub-sampling with given / seeded indices, handle mask, noise.
Run in the build container only:  python oracle/make_golden_dataset.py"""
import importlib.util
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nsdp_amd import synth  # noqa: E402

sys.modules.setdefault("trimesh", types.ModuleType("trimesh"))
spec = importlib.util.spec_from_file_location("ref_dataset_utils", "/root/reference/dataset/utils.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

cfg = {"data": {"num_surf_samples": 300, "num_space_samples": 200, "partial_range": 0.1, "noise_level": 0.02}}
nf, mf = 1000, 700
cano = synth.uniform(11, "cano", (nf, 3), -0.5, 0.5)
src = synth.uniform(11, "src", (nf, 3), -0.5, 0.5)
tgt = synth.uniform(11, "tgt", (nf, 3), -0.5, 0.5)
sp = [synth.uniform(11, "sp%d" % i, (mf, 3), -0.5, 0.5) for i in range(3)]
np.random.seed(123)
c, s, t, idxs = ref.subsample_surface_flow(cfg, cano, src, tgt)
mask = ref.cano_sample_handle_mask(cfg, c, cano.min(axis=0), cano.max(axis=0))
np.random.seed(321)
s_noise = ref.add_noise_to_src(cfg, s)
np.random.seed(55)
sc, ss, st = ref.subsample_space_flow(cfg, sp[0], sp[1], sp[2])
# the partial-shape branch (dataset/utils.py:79-101): the reference's function on the sub-sampled source samples, two ratios; the
# seeds it drew are recorded by replaying the same first RNG call
partial = {}
for tag, ratio, rs in (("a", 0.8, 7), ("b", 0.55, 8)):
    pcfg = {"data": {"partial_shape_ratio": ratio}}
    np.random.seed(rs)
    remain = ref.create_partial_src(pcfg, s, mask)
    np.random.seed(rs)
    choice = np.random.permutation(int((~mask).sum()))[:5]
    partial[f"partial_{tag}_ratio"] = np.float64(ratio)
    partial[f"partial_{tag}_seed_choice"] = choice
    partial[f"partial_{tag}_remain"] = np.asarray(remain, dtype=np.int64)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "dataset_contract.npz"),
                    cano=cano, src=src, tgt=tgt, sp0=sp[0], sp1=sp[1], sp2=sp[2], idxs=idxs, sub_cano=c, sub_src=s,
                    sub_tgt=t, mask=mask, src_noise=s_noise, sc=sc, ss=ss, st=st, **partial)
print("wrote tests/golden/dataset_contract.npz; handle fraction", mask.mean())
