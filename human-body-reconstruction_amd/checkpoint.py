"""Checkpoint / bounds interop with the reference's files.

The reference writes, every `len(loader)//100` steps (train_hash2.py:299-300):
    torch.save(nerf.state_dict(),    f'{model_name}_Nerf_hash.pth')     keys `module.sig_model.{0,2,4}.{weight,bias}`, ...
    torch.save(encoder.state_dict(), f'{model_name}_encoder_hash.pth')  keys `Embedding_list.{0..15}.weight`
and once `np.save('bounds_model.npy', stack([min_bound, max_bound]))` (:115).  Loaders: train_hash2.py:129-133,
nerf2mesh.py:28-29,59-62.  Optimiser / scheduler state is not saved by the reference; `save_trainer_state` adds it
as a separate, optional file so that reference-format files stay byte-compatible in content.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import numpy as np
import torch
import torch.nn as nn


def _with_module_prefix(sd):
    return {("module." + k if not k.startswith("module.") else k): v for k, v in sd.items()}


def _strip_module_prefix(sd):
    return {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}


def save_checkpoint(model_name: str, nerf: nn.Module, encoder: nn.Module, directory: str = ".") -> Tuple[str, str]:
    """Write `{model_name}_Nerf_hash.pth` (always with the DataParallel `module.` prefix, as the reference's files
    have) and `{model_name}_encoder_hash.pth`."""
    p_nerf = os.path.join(directory, f"{model_name}_Nerf_hash.pth")
    p_enc = os.path.join(directory, f"{model_name}_encoder_hash.pth")
    sd = {k: v.detach().cpu().clone() for k, v in nerf.state_dict().items()}
    torch.save(_with_module_prefix(sd), p_nerf)
    torch.save({k: v.detach().cpu().clone() for k, v in encoder.state_dict().items()}, p_enc)
    return p_nerf, p_enc


def load_checkpoint(ckpt_name: str, nerf: nn.Module, encoder: nn.Module, directory: str = ".") -> None:
    """train_hash2.py:129-133.  Accepts files written by the reference or by save_checkpoint; tensors only
    (`weights_only=True`: nothing from the file is executed)."""
    sd_n = torch.load(os.path.join(directory, f"{ckpt_name}_Nerf_hash.pth"), map_location="cpu", weights_only=True)
    sd_e = torch.load(os.path.join(directory, f"{ckpt_name}_encoder_hash.pth"), map_location="cpu", weights_only=True)
    wrapped = isinstance(nerf, (nn.DataParallel, nn.parallel.DistributedDataParallel))
    nerf.load_state_dict(_with_module_prefix(sd_n) if wrapped else _strip_module_prefix(sd_n))
    encoder.load_state_dict(sd_e)


def save_bounds(min_bound: torch.Tensor, max_bound: torch.Tensor, path: str = "bounds_model.npy") -> None:
    """train_hash2.py:115: rows = [min_bound, max_bound]."""
    np.save(path, torch.stack([min_bound.detach().cpu(), max_bound.detach().cpu()]).numpy())


def load_bounds(path: str = "bounds_model.npy"):
    """nerf2mesh.py:28-29: returns (min_bound, max_bound, mu, sigma) with mu = min corner and sigma = bbox diagonal
    (train_hash2.py:117-119)."""
    b = torch.from_numpy(np.load(path, allow_pickle=False)).float()
    mn, mx = b[0], b[1]
    return mn, mx, mn, ((mx - mn) ** 2).sum().sqrt()


def save_trainer_state(path: str, trainer) -> None:
    torch.save({"step": trainer.step_count, "m": trainer.m.cpu(), "v": trainer.v.cpu()}, path)


def load_trainer_state(path: str, trainer) -> None:
    st = torch.load(path, map_location="cpu", weights_only=True)
    trainer.step_count = int(st["step"])
    trainer.m.copy_(st["m"]); trainer.v.copy_(st["v"])
