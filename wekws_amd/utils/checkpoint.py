"""Checkpoint I/O with the reference's on-disk contract (wekws/utils/checkpoint.py:23-57): ``<name>.pt`` holds a plain
``state_dict`` (torch.save), ``<name>.yaml`` next to it holds free-form infos (epoch, lr, cv_loss).  Because
wekws_amd.model.kws_model.KWSModel exposes the reference's state_dict key names, checkpoints written by the reference's
training code load unchanged."""
from __future__ import annotations

import os
import re

import torch
import yaml


def load_checkpoint(model: torch.nn.Module, path: str) -> dict:
    state = torch.load(path, map_location="cpu")
    model.load_state_dict(state)
    info_path = re.sub(r"\.pt$", ".yaml", path)
    if os.path.exists(info_path):
        with open(info_path) as f:
            return yaml.load(f, Loader=yaml.FullLoader) or {}
    return {}


def save_checkpoint(model: torch.nn.Module, path: str, infos=None) -> None:
    torch.save(model.state_dict(), path)
    with open(re.sub(r"\.pt$", ".yaml", path), "w") as f:
        f.write(yaml.dump(infos or {}))
