"""Training harness around the MI355X TDNet path with the loop semantics of the reference's train.py (SURVEY.md
section 8 next-1): seeding, epoch-wise step learning rate, checkpoint cadence, validation cadence and best-model
tracking are the reference's; the model, its train/validate functions and the optimizer come from
``nsdp_amd.model``.  The data loader is whatever yields the reference's ``data_dict`` batches; ``--synthetic`` uses
procedural batches (the dataset readers are outside this round's scope).

    python -m nsdp_amd.train config.yaml experiment_dir [--synthetic 4] [--epochs 2]
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np
import torch

from .checkpoints import load_best_checkpoints, load_checkpoints, save_best_checkpoints, save_checkpoints
from .model import build_model, optimizer_factory
from .model.learningrate import adjust_learning_rate


def seed_everything(seed: int):
    """train.py:68-72: numpy seeds torch, which seeds every GPU."""
    np.random.seed(seed)
    torch.manual_seed(np.random.randint(np.iinfo(np.int32).max))
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(np.random.randint(np.iinfo(np.int32).max))


class _Average:
    """Running mean of the batch losses of an epoch (utils/logger.py AverageAggregator)."""

    def __init__(self):
        self.total, self.count = 0.0, 0

    def add(self, v):
        self.total += float(v)
        self.count += 1

    @property
    def value(self):
        return self.total / max(self.count, 1)


def fit(model, fns, lr_scheduler, optimizer, train_loader, val_loader, config, experiment_directory, args, device,
        log=print):
    """train.py:150-225.  ``fns`` = (train_on_batch, validate_on_batch) from build_model; ``args`` carries
    ``continue_from_epoch`` and ``best_val_loss`` (both updated by the checkpoint loaders and by this loop)."""
    train_on_batch, validate_on_batch = fns
    load_best_checkpoints(model, experiment_directory, args, device)   # best first, then the latest (train.py:153-156)
    load_checkpoints(model, optimizer, experiment_directory, args, device)
    epochs = config["training"].get("epochs", 1000)
    save_every = config["training"].get("save_frequency", 20)
    val_every = config["validation"].get("frequency", 10)
    history = []
    for i in range(args.continue_from_epoch, epochs):
        adjust_learning_rate(lr_scheduler, optimizer, i)
        model.train()
        avg = _Average()
        for b, sample in enumerate(train_loader):
            sample = {k: v.to(device) for k, v in sample.items()}
            avg.add(train_on_batch(model, optimizer, sample, config))
        log("epoch: {} - batches: {} - loss: {:.5f}".format(i + 1, avg.count, avg.value))
        history.append(("train", i, avg.value))
        if (i % save_every) == 0:
            save_checkpoints(i, model, optimizer, experiment_directory)
        if i % val_every == 0 and i > 0:
            model.eval()
            vavg = _Average()
            for b, sample in enumerate(val_loader):
                sample = {k: v.to(device) for k, v in sample.items()}
                vavg.add(validate_on_batch(model, sample, config))
            log("validation epoch: {} - loss: {:.5f}".format(i + 1, vavg.value))
            history.append(("val", i, vavg.value))
            if vavg.value < args.best_val_loss:
                save_best_checkpoints(i, model, experiment_directory, vavg.value)
                args.best_val_loss = vavg.value
    return history


class SyntheticLoader:
    """``n_batches`` procedural data_dict batches per epoch (same generator as bench.py and the parity fixtures)."""

    def __init__(self, seed, n_batches, batch, n_surf=2048, n_query=8192):
        from . import synth
        self.batches = [{k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in
                         synth.make_batch(seed + i, batch, n_surf, n_query).items()} for i in range(n_batches)]

    def __iter__(self):
        return iter(self.batches)

    def __len__(self):
        return len(self.batches)


def initial_weight_files(config, args=None):
    """train.py:140-146: the three weight files that initialise the networks come from the config's ``training``
    section (``weight_file``, ``weight_forward_file``, ``weight_backward_file``); a command-line value, where given,
    overrides the config's.  Returns (weight_file, weight_forward_file, weight_backward_file)."""
    out = []
    for key in ("weight_file", "weight_forward_file", "weight_backward_file"):
        v = config.get("training", {}).get(key, None)
        cli = getattr(args, key, None) if args is not None else None
        out.append(cli if cli is not None else v)
    return tuple(out)


def main(argv=None):
    import yaml
    ap = argparse.ArgumentParser(description="Train a deformation network on MI355X")
    ap.add_argument("config_file")
    ap.add_argument("experiment_directory")
    ap.add_argument("--weight_file", default=None, help="overrides training.weight_file of the config")
    ap.add_argument("--weight_forward_file", default=None, help="overrides training.weight_forward_file")
    ap.add_argument("--weight_backward_file", default=None, help="overrides training.weight_backward_file")
    ap.add_argument("--continue_from_epoch", default=0, type=int)
    ap.add_argument("--seed", type=int, default=27)
    ap.add_argument("--synthetic", type=int, default=4, help="procedural batches per epoch (no dataset readers yet)")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--epochs", type=int, default=None)
    ap.add_argument("--graph", action="store_true",
                    help="capture the train step once and replay it (nsdp_amd.graph_step.GraphedTrainOnBatch): batches of "
                         "one fixed shape replay, others run eagerly; the sequence of optimizer steps is the eager loop's")
    args = ap.parse_args(argv)
    args.best_val_loss = float("inf")
    seed_everything(args.seed)
    device = torch.device("cuda:0")
    with open(args.config_file) as f:
        config = yaml.safe_load(f)
    if args.epochs is not None:
        config["training"]["epochs"] = args.epochs
    os.makedirs(args.experiment_directory, exist_ok=True)
    weights = initial_weight_files(config, args)
    for key, path in zip(("weight_file", "weight_forward_file", "weight_backward_file"), weights):
        if path is not None:
            print("initialising from {} = {}".format(key, path))
    if config["model"]["type"] == "arbitrary" and weights[1] is None and weights[2] is None and weights[0] is None:
        print("WARNING: FlowArbitrary starts from random weights (no weight_forward_file / weight_backward_file in the "
              "config's training section or on the command line)")
    model, train_fn, val_fn, _ = build_model(config, *weights, device=device)
    lr_scheduler, optimizer = optimizer_factory(config["training"], model.parameters())
    if args.graph:
        from .graph_step import GraphedTrainOnBatch
        train_fn = GraphedTrainOnBatch(train_fn)
    train = SyntheticLoader(args.seed, args.synthetic, args.batch)
    val = SyntheticLoader(args.seed + 10000, max(1, args.synthetic // 4), args.batch)
    fit(model, (train_fn, val_fn), lr_scheduler, optimizer, train, val, config, args.experiment_directory, args, device)


if __name__ == "__main__":
    main(sys.argv[1:])
