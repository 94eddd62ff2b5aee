"""Generates tests/golden/eval_metric.npz by running the REFERENCE's utils/eval_metric.py (imported from
/root/reference, with the absent `trimesh` package stubbed -- the three metric functions never touch it) on
procedural inputs.  Run in the build container only:  python oracle/make_golden_eval.py"""
import importlib.util
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nsdp_amd import synth  # noqa: E402

sys.modules.setdefault("trimesh", types.ModuleType("trimesh"))
spec = importlib.util.spec_from_file_location("ref_eval_metric", "/root/reference/utils/eval_metric.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

out = {}
for case, (n, m) in enumerate([(1000, 1000), (3000, 2500), (1, 7)]):
    a = synth.uniform(77 + case, "pts_a", (n, 3), -0.5, 0.5).astype(np.float64)
    b = synth.uniform(77 + case, "pts_b", (m, 3), -0.5, 0.5).astype(np.float64)
    out[f"c{case}_a"], out[f"c{case}_b"] = a, b
    out[f"c{case}_chamfer"] = np.float64(ref.chamfer_distance(a, b))
    k = min(n, m)
    out[f"c{case}_l2"] = np.float64(ref.compute_dist_square(a[:k], b[:k]))
    na = synth.normal(77 + case, "nrm_a", (k, 3)).astype(np.float64)
    nb = synth.normal(77 + case, "nrm_b", (k, 3)).astype(np.float64)
    out[f"c{case}_na"], out[f"c{case}_nb"] = na, nb
    out[f"c{case}_fnc"] = np.float64(ref.normal_consistency(na, nb))
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "eval_metric.npz"), **out)
print("wrote tests/golden/eval_metric.npz", {k: float(v) for k, v in out.items() if np.ndim(v) == 0})
