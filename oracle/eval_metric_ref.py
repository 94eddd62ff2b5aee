"""TEST INFRASTRUCTURE ONLY (oracle): CPU restatement of the reference's dense-inference metrics,
utils/eval_metric.py of tangjiapeng/NSDP.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import it.

Pinned: oracle/make_golden_eval.py imports the reference module itself (with `trimesh`, absent in this image, stubbed
out -- the three functions below do not use it) and stores its outputs in tests/golden/eval_metric.npz;
tests/test_oracle_golden.py checks this restatement against those vectors.
"""
import numpy as np
from scipy.spatial import KDTree


def compute_dist_square(vertices, vertices_gt):
    """utils/eval_metric.py:6-8: mean over vertices of the squared Euclidean distance."""
    return ((vertices - vertices_gt) ** 2).sum(-1).mean()


def normal_consistency(normals_src, normals_tgt):
    """utils/eval_metric.py:11-21: mean |cos| between corresponding (re-normalised) normals."""
    a = normals_src / np.linalg.norm(normals_src, axis=-1, keepdims=True)
    b = normals_tgt / np.linalg.norm(normals_tgt, axis=-1, keepdims=True)
    return np.abs((a * b).sum(axis=-1)).mean()


def chamfer_distance(points, points_gt):
    """utils/eval_metric.py:23-30: symmetric Chamfer-L1 = mean of the two mean nearest-neighbour distances."""
    completeness, _ = KDTree(points_gt).query(points)
    accuracy, _ = KDTree(points).query(points_gt)
    return 0.5 * (accuracy.mean() + completeness.mean())


def face_normals(verts, faces):
    """Unit face normals as trimesh.Trimesh(...).face_normals computes them (cross product of the two edge vectors
    from vertex 0, normalised; utils/eval_metric.py:47-48 reads that attribute)."""
    v = verts[faces]
    n = np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0])
    return n / np.maximum(np.linalg.norm(n, axis=-1, keepdims=True), 1e-30)
