"""Bit-group (de)factorisation of K-bit tokens (reference factorization.py:7-46): group g holds bits
[g*K/m, (g+1)*K/m) of the code, group 0 = low bits."""
from __future__ import annotations

import math

import torch


def _group_bits(codebook_size: int, splits: int) -> int:
    return int(math.log2(codebook_size)) // splits


def combine_factorized_tokens(tokens: torch.Tensor, codebook_size: int, splits: int) -> torch.Tensor:
    """[b, n, m] group indices -> [b, n] codes.  Returns float32 like the reference (exact for K <= 24)."""
    gb = _group_bits(codebook_size, splits)
    shifts = torch.arange(splits, device=tokens.device) * gb
    return (tokens.long() << shifts).sum(-1).to(torch.float32)


def split_factorized_tokens(tokens: torch.Tensor, codebook_size: int, splits: int) -> torch.Tensor:
    """[b, n] codes -> [b, n, m] group indices."""
    gb = _group_bits(codebook_size, splits)
    shifts = torch.arange(splits, device=tokens.device) * gb
    return (tokens.long().unsqueeze(-1) >> shifts) & ((1 << gb) - 1)
