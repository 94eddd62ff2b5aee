"""GPU: dense-inference metrics (HIP kNN Chamfer) against the oracle / the reference's vectors, and a short real
training run through the harness (checkpoints written, loss finite and decreasing on a fixed batch)."""
import argparse
import os

import numpy as np
import pytest
import torch

from helpers import model_cfg
from nsdp_amd import eval_metric, synth, train
from oracle import eval_metric_ref

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def test_metrics_match_reference_vectors():
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "eval_metric.npz"))
    for c in range(3):
        a, b = fx[f"c{c}_a"], fx[f"c{c}_b"]
        ta, tb = torch.from_numpy(a).float().to(DEV), torch.from_numpy(b).float().to(DEV)
        k = min(len(a), len(b))
        # fp32 coordinates and distances against the reference's float64 KD-tree: relative 2e-6
        assert abs(float(eval_metric.chamfer_distance(ta, tb)) - fx[f"c{c}_chamfer"]) <= 2e-6 * fx[f"c{c}_chamfer"]
        assert abs(float(eval_metric.compute_dist_square(ta[:k], tb[:k])) - fx[f"c{c}_l2"]) <= 2e-6 * fx[f"c{c}_l2"]
        na, nb = torch.from_numpy(fx[f"c{c}_na"]).float().to(DEV), torch.from_numpy(fx[f"c{c}_nb"]).float().to(DEV)
        assert abs(float(eval_metric.normal_consistency(na, nb)) - fx[f"c{c}_fnc"]) <= 2e-6


def test_chamfer_at_evaluation_size_and_nn_indices():
    """30 000 x 30 000 points (the reference's point-cloud size): nearest-neighbour distances equal the KD-tree's."""
    a = synth.uniform(5, "a", (30000, 3), -0.5, 0.5)
    b = synth.uniform(5, "b", (30000, 3), -0.5, 0.5)
    d = eval_metric.nn_distance(torch.from_numpy(a).to(DEV), torch.from_numpy(b).to(DEV)).cpu().numpy()
    from scipy.spatial import KDTree
    ref, _ = KDTree(b.astype(np.float64)).query(a.astype(np.float64))
    np.testing.assert_allclose(d, ref, rtol=2e-6, atol=1e-7)
    cd = float(eval_metric.chamfer_distance(torch.from_numpy(a).to(DEV), torch.from_numpy(b).to(DEV)))
    assert abs(cd - eval_metric_ref.chamfer_distance(a.astype(np.float64), b.astype(np.float64))) <= 2e-6 * cd


def test_compute_evaluation_metrics_on_a_mesh():
    g = np.random.RandomState(0)
    verts = g.rand(500, 3).astype(np.float32)
    faces = g.randint(0, 500, size=(900, 3)).astype(np.int64)
    faces = faces[(faces[:, 0] != faces[:, 1]) & (faces[:, 1] != faces[:, 2]) & (faces[:, 0] != faces[:, 2])]
    pred = verts + 0.01 * g.randn(500, 3).astype(np.float32)
    out = {"verts_tgt_pred": torch.from_numpy(pred).to(DEV)[None], "verts_tgt": torch.from_numpy(verts)[None],
           "faces": torch.from_numpy(faces)[None]}
    gen = torch.Generator(device=DEV).manual_seed(1)
    m = eval_metric.compute_evaluation_metrics(out, pointcloud_size=20000, generator=gen)
    assert abs(m["l2"] - eval_metric_ref.compute_dist_square(pred.astype(np.float64), verts.astype(np.float64))) < 1e-8
    fn = eval_metric_ref.normal_consistency(eval_metric_ref.face_normals(pred.astype(np.float64), faces),
                                            eval_metric_ref.face_normals(verts.astype(np.float64), faces))
    assert abs(m["fnc"] - fn) < 1e-5
    assert 0.0 < m["cd"] < 0.05          # sampled surfaces 1 cm apart (statistical: own sampling RNG)


def test_short_training_run_through_the_harness(tmp_path):
    from nsdp_amd.model import build_model, optimizer_factory
    cfg = model_cfg("forward", [2048, 500, 100])
    cfg["training"] = {"epochs": 3, "save_frequency": 1, "optimizer": "Adam", "lr": 5e-4, "lr_step": 200,
                       "lr_decay": 0.1, "weight_decay": 0.0}
    cfg["validation"] = {"frequency": 1}
    train.seed_everything(27)
    model, train_fn, val_fn, _ = build_model(cfg, device=DEV)
    sched, opt = optimizer_factory(cfg["training"], model.parameters())
    loader = train.SyntheticLoader(3, 2, 2, n_surf=2048, n_query=1024)
    args = argparse.Namespace(continue_from_epoch=0, best_val_loss=float("inf"))
    hist = train.fit(model, (train_fn, val_fn), sched, opt, loader, loader, cfg, str(tmp_path), args, DEV,
                     log=lambda *_: None)
    losses = [h[2] for h in hist if h[0] == "train"]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    files = os.listdir(str(tmp_path))
    assert {"model_00000", "model_00001", "model_00002", "opt_00002"} <= set(files)
    assert any(f.startswith("modelbest_") for f in files)
    # the saved state dict has the reference's keys (loadable by either code base)
    sd = torch.load(os.path.join(str(tmp_path), "model_00002"), map_location="cpu")
    assert "decoder.ct1.fc_gamma.0.weight" in sd and "encoder.enc_sdf.weight" in sd


def test_harness_with_the_replayed_train_step(tmp_path):
    """fit() with train_on_batch replaced by its graph-replay drop-in (GraphedTrainOnBatch): same loop, same checkpoints,
    the learning-rate schedule reaching the captured step through the device-side rate, a batch of another shape falling back
    to the eager function."""
    from helpers import nondeterministic_knobs
    if nondeterministic_knobs():
        pytest.skip("the step is not bit-reproducible under " + ", ".join(nondeterministic_knobs()))
    from nsdp_amd.graph_step import GraphedTrainOnBatch
    from nsdp_amd.model import build_model, optimizer_factory
    cfg = model_cfg("forward", [256, 64, 16])
    cfg["training"] = {"epochs": 4, "save_frequency": 2, "optimizer": "Adam", "lr": 5e-4, "lr_step": 2,
                       "lr_decay": 0.1, "weight_decay": 0.0}
    cfg["validation"] = {"frequency": 2}
    hists = []
    for graph in (False, True):
        train.seed_everything(27)
        model, train_fn, val_fn, _ = build_model(cfg, device=DEV)
        sched, opt = optimizer_factory(cfg["training"], model.parameters())
        loader = train.SyntheticLoader(3, 3, 2, n_surf=256, n_query=128)
        odd = train.SyntheticLoader(9, 1, 1, n_surf=256, n_query=128)          # one batch of another shape per epoch
        both = list(loader) + list(odd)
        from nsdp_amd.graph_step import capturable_adam
        capturable_adam(opt)        # (both runs on the fused capturable Adam: the same arithmetic, so the runs can be held EQUAL)
        fn = GraphedTrainOnBatch(train_fn) if graph else train_fn
        args = argparse.Namespace(continue_from_epoch=0, best_val_loss=float("inf"))
        sub = tmp_path / ("g" if graph else "e")
        sub.mkdir()
        hists.append(train.fit(model, (fn, val_fn), sched, opt, both, loader, cfg, str(sub), args, DEV, log=lambda *_: None))
        if graph:
            assert fn.replays == 4 * 3 - 1 and fn.eager_calls == 4 + 1      # (the very first step is eager)
            assert torch.is_tensor(opt.param_groups[0]["lr"]) and abs(float(opt.param_groups[0]["lr"]) - 5e-5) < 1e-9
    e = [h[2] for h in hists[0] if h[0] == "train"]
    g = [h[2] for h in hists[1] if h[0] == "train"]
    assert len(e) == len(g) == 4
    # the step is deterministic (no floating-point atomics) and a replay reproduces the eager step bit for bit: the epoch
    # losses of the two runs -- replays, the validation passes between them, the odd-shape eager batch -- are EQUAL, not close
    assert e == g, (e, g)
    ve, vg = [h[2] for h in hists[0] if h[0] == "val"], [h[2] for h in hists[1] if h[0] == "val"]
    assert ve == vg and len(ve) >= 1, (ve, vg)
    assert g[-1] < g[0]


@pytest.mark.parametrize("inverse,noise_level", [(False, 0.0), (True, 0.02)])
def test_prepare_batch_matches_oracle(inverse, noise_level):
    from nsdp_amd import dataset
    from oracle import dataset_ref
    B, nf, mf = 3, 1500, 900
    cfg = {"arbitrary": False, "inverse": inverse, "num_surf_samples": 512, "num_space_samples": 400,
           "partial_range": 0.1, "noise_level": noise_level, "partial_shape_ratio": 1.0}
    mk = lambda tag, n: np.stack([synth.uniform(40 + b, tag, (n, 3), -0.5, 0.5) for b in range(B)])
    raw = {r: {"surface_samples": mk(r + "s", nf), "surface_normals": mk(r + "n", nf), "space_samples": mk(r + "q", mf)}
           for r in ("cano", "src", "tgt")}
    dev = {r: {k: torch.from_numpy(v).to(DEV) for k, v in d.items()} for r, d in raw.items()}
    gen = torch.Generator(device=DEV).manual_seed(7)
    surf_idx = dataset.random_subset(B, nf, 512, DEV, gen)
    space_idx = dataset.random_subset(B, mf, 400, DEV, gen)
    noise = torch.randn(B, 512, 3, device=DEV, generator=gen)
    out = dataset.prepare_batch(cfg, dev["cano"], dev["src"], dev["tgt"], surf_idx, space_idx, noise)
    assert out["surface_samples_inputs"].shape == (B, 512, 7) and out["space_samples_src"].shape == (B, 400, 3)
    for b in range(B):
        pick = lambda r: {k: v[b] for k, v in raw[r].items()}
        ref = dataset_ref.sample_contract(cfg, pick("cano"), pick("src"), pick("tgt"),
                                          surf_idxs=surf_idx[b].cpu().numpy(), noise=noise[b].cpu().numpy())
        for key in ("surface_samples_cano", "surface_samples_tgt", "surface_normals_src", "surface_samples_inputs"):
            np.testing.assert_allclose(out[key][b].cpu().numpy(), ref[key], rtol=0, atol=1e-7, err_msg=key)
        assert np.array_equal(out["cano_handle_sample_idx"][b].cpu().numpy(), ref["cano_handle_sample_idx"])
        sp = space_idx[b].cpu().numpy()
        src_space = raw["tgt" if inverse else "src"]["space_samples"][b][sp]
        np.testing.assert_array_equal(out["space_samples_src"][b].cpu().numpy(), src_space)
    # every index set is a permutation prefix: unique rows
    assert all(len(set(surf_idx[b].tolist())) == 512 for b in range(B))


def test_create_partial_src_matches_the_reference_vectors_and_prepare_batch_carves_the_same_holes():
    """dataset/utils.py:79-101 (the partial-shape branch of the data contract) on the device against the outputs of the IMPORTED
    reference function (tests/golden/dataset_contract.npz, oracle/make_golden_dataset.py: two ratios, the seeds the reference
    drew supplied as positions within the non-handle samples), then through prepare_batch at batch 1: every per-sample array
    keeps exactly the reference's remaining rows.  Without given seeds: seeds are non-handle samples, holes have the size
    the reference computes, and a batch is deterministic under a seeded generator."""
    import os
    from nsdp_amd import dataset
    from oracle import dataset_ref
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "dataset_contract.npz"))
    src = torch.from_numpy(fx["sub_src"]).to(DEV)[None]
    mask = torch.from_numpy(fx["mask"]).to(DEV)[None]
    for tag in "ab":
        ratio, choice, want = float(fx[f"partial_{tag}_ratio"]), fx[f"partial_{tag}_seed_choice"], fx[f"partial_{tag}_remain"]
        keep = dataset.create_partial_src(ratio, src, mask, seed_choice=choice[None])
        assert np.array_equal(keep[0].nonzero()[:, 0].cpu().numpy(), want)
        assert np.array_equal(dataset_ref.create_partial_src(ratio, fx["sub_src"], fx["mask"], seed_choice=choice), want)
    # two samples at once, different seeds per sample
    both = dataset.create_partial_src(0.8, src.expand(2, -1, -1).contiguous(), mask.expand(2, -1).contiguous(),
                                      seed_choice=np.stack([fx["partial_a_seed_choice"], fx["partial_a_seed_choice"][::-1]]))
    assert torch.equal(both[0], both[1])                       # (the same seeds in another order: the same holes)
    # through prepare_batch (batch 1): rows of every array = the reference's remain_idx
    cfg = {"arbitrary": False, "inverse": False, "num_surf_samples": 300, "num_space_samples": 10 ** 9, "partial_range": 0.1,
           "noise_level": 0.0, "partial_shape_ratio": 0.8}
    d = lambda a: {"surface_samples": torch.from_numpy(fx[a]).to(DEV)[None], "surface_normals": torch.from_numpy(fx[a]).to(DEV)[None],
                   "space_samples": torch.from_numpy(fx["sp0"]).to(DEV)[None]}
    idx = torch.from_numpy(fx["idxs"].astype(np.int32)).to(DEV)[None]
    out = dataset.prepare_batch(cfg, d("cano"), d("src"), d("tgt"), surf_idx=idx, partial_seed_choice=fx["partial_a_seed_choice"][None])
    want = fx["partial_a_remain"]
    assert np.array_equal(out["partial_remain_idx"][0].cpu().numpy(), want)
    np.testing.assert_array_equal(out["surface_samples_src"][0].cpu().numpy(), fx["sub_src"][want])
    np.testing.assert_array_equal(out["surface_samples_cano"][0].cpu().numpy(), fx["sub_cano"][want])
    np.testing.assert_array_equal(out["cano_handle_sample_idx"][0, :, 0].cpu().numpy(), fx["mask"][want])
    inputs = np.concatenate([fx["sub_src"], fx["sub_tgt"] * fx["mask"][:, None], fx["mask"][:, None]], axis=1).astype(np.float32)
    np.testing.assert_array_equal(out["surface_samples_inputs"][0].cpu().numpy(), inputs[want])
    # random seeds: non-handle seeds, hole size, reproducible
    g = torch.Generator(device=DEV).manual_seed(3)
    big = torch.rand(4, 2048, 3, device=DEV, generator=g) - 0.5
    hmask = big[..., 1] > 0.3
    k1 = dataset.create_partial_src(0.8, big, hmask, generator=torch.Generator(device=DEV).manual_seed(9))
    k2 = dataset.create_partial_src(0.8, big, hmask, generator=torch.Generator(device=DEV).manual_seed(9))
    assert torch.equal(k1, k2)
    per_hole = int(0.2 * 2048 // 5)
    removed = (~k1).sum(dim=1)
    assert bool(((removed >= per_hole) & (removed <= 5 * per_hole)).all()), removed
    assert torch.equal(dataset.create_partial_src(1.0, big, hmask), torch.ones_like(hmask))


@pytest.mark.gpu
def test_harness_with_the_next_batch_s_geometry_pipelined_under_the_step(tmp_path):
    """GraphedTrainOnBatch(pipeline_geometry=): every replay takes its index sets (FPS, kNN, inverse lists) from
    graph_step.PipelinedGeometry and computes the NEXT batch's on a stream of its own; fit() looks one batch ahead.  The searches
    are the same searches -- only when they run changes -- so the loop is held EQUAL, epoch loss by epoch loss, to the eager
    loop: across epoch boundaries (no next batch announced: the step primes eagerly), odd-shape eager batches and validation
    passes in between."""
    from helpers import nondeterministic_knobs
    if nondeterministic_knobs():
        pytest.skip("the step is not bit-reproducible under " + ", ".join(nondeterministic_knobs()))
    from nsdp_amd.graph_step import GraphedTrainOnBatch, capturable_adam
    from nsdp_amd.model import build_model, optimizer_factory
    cfg = model_cfg("forward", [256, 64, 16])
    cfg["training"] = {"epochs": 4, "save_frequency": 2, "optimizer": "Adam", "lr": 5e-4, "lr_step": 2,
                       "lr_decay": 0.1, "weight_decay": 0.0}
    cfg["validation"] = {"frequency": 2}
    hists, fns = [], []
    for mode in ("eager", "piped"):
        train.seed_everything(27)
        model, train_fn, val_fn, _ = build_model(cfg, device=DEV)
        sched, opt = optimizer_factory(cfg["training"], model.parameters())
        loader = train.SyntheticLoader(3, 4, 2, n_surf=256, n_query=128)
        odd = train.SyntheticLoader(9, 1, 1, n_surf=256, n_query=128)
        both = list(loader)[:2] + list(odd) + list(loader)[2:]          # the odd-shape batch in the MIDDLE of every epoch
        capturable_adam(opt)
        fn = train_fn if mode == "eager" else GraphedTrainOnBatch(
            train_fn, pipeline_geometry=lambda d: (d["space_samples_src"], d["surface_samples_inputs"]))
        args = argparse.Namespace(continue_from_epoch=0, best_val_loss=float("inf"))
        sub = tmp_path / mode
        sub.mkdir()
        hists.append(train.fit(model, (fn, val_fn), sched, opt, both, loader, cfg, str(sub), args, DEV, log=lambda *_: None))
        fns.append(fn)
    fn = fns[1]
    assert fn.accepts_next_batch and fn._pipe is not None
    assert fn.replays == 4 * 4 - 1 and fn.eager_calls == 4 + 1
    # announced: batch 2 of an epoch (after batch 1) and batch 4 (after batch 3, announced through the odd one? no: the odd batch
    # is announced to batch 2's replay and does not match the static shapes) -- what matters is that every unannounced batch primed
    assert 1 <= fn.unannounced <= fn.replays
    e = [h[2] for h in hists[0] if h[0] == "train"]
    g = [h[2] for h in hists[1] if h[0] == "train"]
    assert len(e) == len(g) == 4 and e == g, (e, g)
    ve, vg = [h[2] for h in hists[0] if h[0] == "val"], [h[2] for h in hists[1] if h[0] == "val"]
    assert ve == vg and len(ve) >= 1, (ve, vg)
