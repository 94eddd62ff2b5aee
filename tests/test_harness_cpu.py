"""Training-harness semantics (reference train.py / utils/checkpoints.py) with a stub model on CPU: file naming,
save / validation cadence, best-model tracking, resume -- and the eval-metric oracle against the vectors produced
by the reference's own utils/eval_metric.py."""
import argparse
import os

import numpy as np
import torch

from nsdp_amd import checkpoints, train
from oracle import eval_metric_ref


class _Stub(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.zeros(3))


def _fns(log):
    def train_on_batch(model, optimizer, sample, config):
        optimizer.zero_grad()
        loss = ((model.w - sample["t"]) ** 2).sum()
        loss.backward()
        optimizer.step()
        log.append(("t", optimizer.param_groups[0]["lr"]))
        return loss.item()

    def validate_on_batch(model, sample, config):
        return ((model.w - sample["t"]) ** 2).sum().item()
    return train_on_batch, validate_on_batch


def _cfg(epochs):
    return {"training": {"epochs": epochs, "save_frequency": 2, "optimizer": "SGD", "lr": 0.1, "lr_step": 3,
                         "lr_decay": 0.5, "momentum": 0.0},
            "validation": {"frequency": 2}}


def test_fit_cadence_naming_and_resume(tmp_path):
    from nsdp_amd.model import optimizer_factory
    d = str(tmp_path)
    cfg = _cfg(5)
    model = _Stub()
    sched, opt = optimizer_factory(cfg["training"], model.parameters())
    loader = [{"t": torch.ones(3)}, {"t": torch.ones(3)}]
    args = argparse.Namespace(continue_from_epoch=0, best_val_loss=float("inf"))
    log = []
    hist = train.fit(model, _fns(log), sched, opt, loader, loader, cfg, d, args, "cpu", log=lambda *_: None)
    files = sorted(os.listdir(d))
    # checkpoints at epochs 0, 2, 4 (i % 2 == 0); validation at 2 and 4 only (i > 0), each one improving -> best files
    assert [f for f in files if f.startswith("model_")] == ["model_00000", "model_00002", "model_00004"]
    assert [f for f in files if f.startswith("opt_")] == ["opt_00000", "opt_00002", "opt_00004"]
    best = [f for f in files if f.startswith("modelbest_")]
    assert [b[:15] for b in best] == ["modelbest_00002", "modelbest_00004"]
    assert [h[1] for h in hist if h[0] == "val"] == [2, 4]
    # epoch-wise step schedule: lr = 0.1 * 0.5 ** (epoch // 3), two batches per epoch
    lrs = [lr for _, lr in log]
    assert lrs == [0.1] * 6 + [0.05] * 4
    # resume: latest pair wins over the best file's epoch, optimizer state restored
    model2 = _Stub()
    sched2, opt2 = optimizer_factory(cfg["training"], model2.parameters())
    args2 = argparse.Namespace(continue_from_epoch=0, best_val_loss=float("inf"))
    cfg7 = _cfg(7)
    hist2 = train.fit(model2, _fns([]), sched2, opt2, loader, loader, cfg7, d, args2, "cpu", log=lambda *_: None)
    assert [h[1] for h in hist2 if h[0] == "train"] == [5, 6]
    assert args2.best_val_loss <= float(best[-1][16:])
    assert "model_00006" in os.listdir(d)


def test_best_checkpoint_name_round_trip(tmp_path):
    d = str(tmp_path)
    m = _Stub()
    checkpoints.save_best_checkpoints(12, m, d, 0.00123456)
    assert os.listdir(d) == ["modelbest_00012_0.001235"]
    args = argparse.Namespace(continue_from_epoch=0, best_val_loss=float("inf"))
    checkpoints.load_best_checkpoints(_Stub(), d, args, "cpu")
    assert args.continue_from_epoch == 13 and abs(args.best_val_loss - 0.001235) < 1e-9


def test_seed_everything_matches_reference_recipe():
    train.seed_everything(27)
    a = torch.rand(3)
    np.random.seed(27)
    torch.manual_seed(np.random.randint(np.iinfo(np.int32).max))
    assert torch.equal(a, torch.rand(3))


def test_eval_metric_oracle_matches_reference_vectors():
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "eval_metric.npz"))
    for c in range(3):
        a, b = fx[f"c{c}_a"], fx[f"c{c}_b"]
        k = min(len(a), len(b))
        assert eval_metric_ref.chamfer_distance(a, b) == fx[f"c{c}_chamfer"]
        assert eval_metric_ref.compute_dist_square(a[:k], b[:k]) == fx[f"c{c}_l2"]
        assert eval_metric_ref.normal_consistency(fx[f"c{c}_na"], fx[f"c{c}_nb"]) == fx[f"c{c}_fnc"]


def test_dataset_contract_oracle_matches_reference_vectors():
    from oracle import dataset_ref
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "dataset_contract.npz"))
    np.random.seed(123)
    c, s, t, idxs = dataset_ref.subsample_surface_flow(300, fx["cano"], fx["src"], fx["tgt"])
    assert np.array_equal(idxs, fx["idxs"]) and np.array_equal(c, fx["sub_cano"]) and np.array_equal(t, fx["sub_tgt"])
    mask = dataset_ref.cano_sample_handle_mask(0.1, c, fx["cano"].min(axis=0), fx["cano"].max(axis=0))
    assert np.array_equal(mask, fx["mask"])
    np.random.seed(55)
    sc, ss, st = dataset_ref.subsample_space_flow(200, fx["sp0"], fx["sp1"], fx["sp2"])
    assert np.array_equal(sc, fx["sc"]) and np.array_equal(ss, fx["ss"]) and np.array_equal(st, fx["st"])
    np.random.seed(321)
    cfg = {"arbitrary": False, "inverse": False, "num_surf_samples": 300, "num_space_samples": 10 ** 9,
           "partial_range": 0.1, "noise_level": 0.02}
    d = lambda a: {"surface_samples": a, "surface_normals": a, "space_samples": fx["sp0"]}
    out = dataset_ref.sample_contract(cfg, d(fx["cano"]), d(fx["src"]), d(fx["tgt"]), surf_idxs=fx["idxs"])
    assert np.array_equal(out["surface_samples_src"], fx["src_noise"])          # same RNG stream as the reference's call
    assert out["surface_samples_inputs"].shape == (300, 7) and out["surface_samples_inputs"].dtype == np.float32
    assert np.array_equal(out["surface_samples_inputs"][:, 6] > 0, fx["mask"])
    # the partial-shape branch (dataset/utils.py:79-101): the restatement against the imported reference function's outputs,
    # with the seeds the reference drew and with the same RNG call
    for tag, rs in (("a", 7), ("b", 8)):
        ratio, want = float(fx[f"partial_{tag}_ratio"]), fx[f"partial_{tag}_remain"]
        got = dataset_ref.create_partial_src(ratio, fx["sub_src"], fx["mask"], seed_choice=fx[f"partial_{tag}_seed_choice"])
        assert np.array_equal(got, want)
        np.random.seed(rs)
        assert np.array_equal(dataset_ref.create_partial_src(ratio, fx["sub_src"], fx["mask"]), want)
    assert np.array_equal(dataset_ref.create_partial_src(1.0, fx["sub_src"], fx["mask"]), np.arange(300))


def test_initial_weight_files_come_from_the_config_like_the_reference():
    """Reference train.py:140-146 reads the three weight files from config['training']; the CLI may override each."""
    cfg = {"training": {"weight_forward_file": "fwd.pt", "weight_backward_file": "bwd.pt"}}
    assert train.initial_weight_files(cfg) == (None, "fwd.pt", "bwd.pt")
    args = argparse.Namespace(weight_file="w.pt", weight_forward_file=None, weight_backward_file="other.pt")
    assert train.initial_weight_files(cfg, args) == ("w.pt", "fwd.pt", "other.pt")
    assert train.initial_weight_files({"training": {}}) == (None, None, None)


def test_main_passes_all_three_weight_files_to_build_model(tmp_path, monkeypatch):
    import yaml
    seen = {}

    def fake_build(config, weight_file=None, weight_forward_file=None, weight_backward_file=None, device="cpu"):
        seen["args"] = (weight_file, weight_forward_file, weight_backward_file)
        raise SystemExit(0)                      # nothing below build_model matters for this test
    monkeypatch.setattr(train, "build_model", fake_build)
    cfg = {"model": {"type": "arbitrary"}, "training": {"weight_forward_file": "F", "weight_backward_file": "B"},
           "validation": {}}
    path = tmp_path / "c.yaml"
    path.write_text(yaml.safe_dump(cfg))
    try:
        train.main([str(path), str(tmp_path / "exp")])
    except SystemExit:
        pass
    assert seen["args"] == (None, "F", "B")


def test_loader_side_sharding_equals_every_world_th_batch():
    """DataParallel.shard: a loader's own shard(rank, world) (SyntheticLoader) and the lazy every-world-th fallback over a plain
    iterable hand every rank the same batches, drop the trailing partial group, and the fallback holds one batch at a time."""
    import types
    from nsdp_amd.parallel import DataParallel
    loader = train.SyntheticLoader(3, 7, 1, n_surf=8, n_query=4)
    model = torch.nn.Linear(2, 2)
    for world in (1, 2, 3):
        for rank in range(world):
            dp = DataParallel(model, rank, world)
            own = list(dp.shard(loader))
            lazy = dp.shard(iter(loader.batches))              # no shard() method: the fallback
            assert isinstance(lazy, types.GeneratorType)
            lazy = list(lazy)
            assert len(own) == len(lazy) == 7 // world
            for a, b in zip(own, lazy):
                assert a is b
            ids = [id(b) for b in loader.batches]
            assert [ids.index(id(a)) for a in own] == [g * world + rank for g in range(7 // world)]


def test_one_batch_look_ahead_and_the_geometry_hand_over_protocol():
    """Host logic of the pipelined-geometry step (graph_step.PipelinedGeometry / GraphedTrainOnBatch, train._with_next): the
    look-ahead pairs every batch with its successor; a bundle's tensors are enumerated once each in a fixed order with the query
    points (an INPUT) left out; an announced batch is recognised by identity AND version."""
    import torch
    from nsdp_amd import train
    from nsdp_amd.graph_step import GraphedTrainOnBatch, _geometry_tensors
    batches = [{"a": torch.full((2,), float(i))} for i in range(4)]
    pairs = list(train._with_next(batches, torch.device("cpu")))
    assert [int(c["a"][0]) for c, _ in pairs] == [0, 1, 2, 3]
    assert [None if n is None else int(n["a"][0]) for _, n in pairs] == [1, 2, 3, None]
    assert all(pairs[i][1] is pairs[i + 1][0] for i in range(3))          # the announced dict IS the next call's batch
    assert list(train._with_next([], torch.device("cpu"))) == []
    t = [torch.zeros(3, dtype=torch.int32) for _ in range(5)]
    q = torch.zeros(2)
    g = {"query_points": q, "query_idx": t[0],
         "encoder": {"levels": [{"fps_idx": t[1], "new_xyz": t[2], "sa_inv": (t[3], t[4]), "blk_idx": None}], "anchors": t[2]}}
    flat = _geometry_tensors(g)
    assert len(flat) == 5 and all(any(f is x for f in flat) for x in t) and not any(f is q for f in flat)
    assert [id(x) for x in _geometry_tensors(g)] == [id(x) for x in flat]          # a fixed order
    d = {"x": torch.zeros(2), "y": torch.ones(2), "tag": "meta"}
    ann = {k: (v, v._version) for k, v in d.items() if torch.is_tensor(v)}
    assert GraphedTrainOnBatch._same_batch(ann, d)
    assert not GraphedTrainOnBatch._same_batch(ann, {"x": d["x"].clone(), "y": d["y"], "tag": "meta"})
    d["x"].add_(1)                                                                   # refilled in place: another batch
    assert not GraphedTrainOnBatch._same_batch(ann, d)
    assert not GraphedTrainOnBatch._same_batch(None, d)
