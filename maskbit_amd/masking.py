"""Mask schedule of the sampler (host-side scalar math; reference masking.py:41-65)."""
from __future__ import annotations

import math

import torch

_MODES = ("linear", "square", "cosine", "arccos", "root")


def get_masking_ratio(progress: float, mode: str = "arccos") -> torch.Tensor:
    """progress in (0, 1] -> fraction of maskable positions to keep masked, as a float32 0-d
    tensor clamped to [1e-6, 1] (the float32 rounding is part of the contract: it decides k)."""
    if mode not in _MODES:
        raise ValueError("Invalid mode. Choose between 'linear','square', 'cosine', 'arccos', 'root'.")
    r = torch.tensor(progress)
    ratio = {
        "root": lambda: 1 - r ** 0.5,
        "square": lambda: 1 - r ** 2,
        "cosine": lambda: torch.cos(r * math.pi * 0.5),
        "arccos": lambda: torch.acos(r) / (math.pi * 0.5),
        "linear": lambda: 1 - r,
    }[mode]()
    return torch.clamp(ratio, 1e-6, 1.0)
