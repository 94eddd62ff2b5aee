"""Checkpoint files of an experiment directory -- same names, contents and resume rules as the reference's
utils/checkpoints.py, so runs started with either code base continue with the other:
``model_%05d`` / ``opt_%05d`` (state dicts, every ``save_frequency`` epochs) and ``modelbest_%05d_%f`` (best
validation loss so far)."""
from __future__ import annotations

import os

import torch


def load_checkpoints(model, optimizer, experiment_directory, args, device):
    """Latest ``model_N`` + ``opt_N`` pair; sets ``args.continue_from_epoch = N + 1`` (utils/checkpoints.py:8-34)."""
    ids = [int(f[6:]) for f in os.listdir(experiment_directory) if f.startswith("model_")]
    if not ids:
        return
    last = max(ids)
    model_path = os.path.join(experiment_directory, "model_{:05d}".format(last))
    opt_path = os.path.join(experiment_directory, "opt_{:05d}".format(last))
    if not (os.path.exists(model_path) and os.path.exists(opt_path)):
        return
    model.load_state_dict(torch.load(model_path, map_location=device))
    optimizer.load_state_dict(torch.load(opt_path, map_location=device))
    args.continue_from_epoch = last + 1


def save_checkpoints(epoch, model, optimizer, experiment_directory):
    """utils/checkpoints.py:37-45."""
    torch.save(model.state_dict(), os.path.join(experiment_directory, "model_{:05d}".format(epoch)))
    torch.save(optimizer.state_dict(), os.path.join(experiment_directory, "opt_{:05d}".format(epoch)))


def load_best_checkpoints(model, experiment_directory, args, device):
    """Lexicographically last ``modelbest_EEEEE_loss`` file; sets ``continue_from_epoch`` and ``best_val_loss``
    (utils/checkpoints.py:50-70)."""
    ids = [f[10:] for f in os.listdir(experiment_directory) if f.startswith("modelbest_")]
    if not ids:
        return
    last = sorted(ids)[-1]
    epoch, val_loss = int(last[0:5]), float(last[6:])
    path = os.path.join(experiment_directory, "modelbest_{:05d}_{:03f}".format(epoch, val_loss))
    if not os.path.exists(path):
        return
    model.load_state_dict(torch.load(path, map_location=device))
    args.continue_from_epoch = epoch + 1
    args.best_val_loss = val_loss


def save_best_checkpoints(epoch, model, experiment_directory, val_loss):
    """utils/checkpoints.py:72-76."""
    torch.save(model.state_dict(),
               os.path.join(experiment_directory, "modelbest_{:05d}_{:03f}".format(epoch, val_loss)))
