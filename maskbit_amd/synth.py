"""Seeded synthetic checkpoints with the reference's key names and shapes.

No checkpoint or dataset can be downloaded here, so benchmarks, tests and the golden fixtures all run on random-init
weights of the real architectures.  This module is the single source of those weights (pure PyTorch CPU, deterministic
from a seed): ``bench.py`` times them, ``tests/`` and ``oracle/`` load the same tensors into the HIP engine, the CPU
oracle and -- in the build container -- the real reference (``oracle/make_golden.py``), and the fixtures under
``tests/golden/`` guard them with sha256 sentinels.  It contains no part of the algorithm.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Tuple

import torch

Tensor = torch.Tensor
StateDict = Dict[str, Tensor]


# --------------------------------------------------------------------------- configs
@dataclass(frozen=True)
class GenCfg:
    """Generator hyper-parameters (bert.py:345-358 constructor arguments)."""
    bits: int = 12            # K = log2(codebook_size)
    splits: int = 2           # m
    hidden: int = 1024        # d
    depth: int = 24           # L
    heads: int = 16           # H
    mlp: int = 4096           # f
    seq: int = 256            # (img_size // input_stride) ** 2
    nclass: int = 1000
    prenorm: bool = False     # use_prenorm (bert.py:49-59,106-123,326-327,498-499)
    kind: str = "lfq"         # "lfq": LFQBert (bit-vector input projection); "bert": Bert (embedding tables, tied output head)

    @property
    def group_bits(self) -> int:
        return self.bits // self.splits

    @property
    def group_codes(self) -> int:          # C, also the mask token id
        return 1 << self.group_bits


@dataclass(frozen=True)
class TokCfg:
    """Tokenizer hyper-parameters (configs/tokenizer/*.yaml, model.vq_model)."""
    token_size: int = 12
    hidden_channels: int = 128
    channel_mult: Tuple[int, ...] = (1, 1, 2, 2, 4)
    num_resolutions: int = 5
    num_res_blocks: int = 2
    num_channels: int = 3
    sample_with_conv: bool = True


def _index_to_bits(idx: Tensor, nbits: int) -> Tensor:
    """LSB-first {-1,+1} expansion of integer codes (the lookup-free codebook buffer, lookup_free.py:96-111)."""
    weights = (1 << torch.arange(nbits, dtype=torch.int64))
    return ((idx.long().unsqueeze(-1) & weights) != 0).to(torch.float32) * 2.0 - 1.0


# --------------------------------------------------------------------------- seeded synthetic weights
def make_generator_weights(cfg: GenCfg, seed: int = 0, head_gain: float = 1.0) -> StateDict:
    """Build-own seeded weights with the reference checkpoint's key names and shapes
    (SURVEY.md 8b).  randn*0.02 for Linear / Embedding / pos_emb, LN gamma=1 beta=0;
    ``head_gain`` scales prediction_layer.weight so the softmax is peaky enough to
    make logit errors visible in the sampled tokens."""
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s: torch.randn(*s, generator=g) * 0.02
    d, f = cfg.hidden, cfg.mlp
    if cfg.kind == "bert":            # Bert (bert.py:222-262): embedding tables instead of the bit projection, tied head + per-position bias
        sd: StateDict = {"pos_emb": rn(1, cfg.seq + 1, d), "class_emb.weight": rn(cfg.nclass + 1, d)}
        for q in range(cfg.splits):
            sd[f"tok_emb_list.{q}.weight"] = rn(cfg.group_codes + 1, d) * (head_gain if head_gain != 1.0 else 1.0)
        sd["first_layer.0.weight"] = torch.ones(d) + rn(d); sd["first_layer.0.bias"] = rn(d)
    else:
        sd = {
            "pos_emb": rn(1, cfg.seq + 1, d),
            "bits_to_indices": (1 << torch.arange(cfg.group_bits)).to(torch.int32),
            "class_emb.weight": rn(cfg.nclass + 1, d),
            "input_proj.weight": rn(d, cfg.bits), "input_proj.bias": rn(d),
            "first_layer.0.weight": torch.ones(d) + rn(d), "first_layer.0.bias": rn(d),
        }
    for l in range(cfg.depth):
        a, ff = f"transformer.layers.{l}.0", f"transformer.layers.{l}.1"
        sd[a + ".mha.in_proj_weight"] = rn(3 * d, d); sd[a + ".mha.in_proj_bias"] = rn(3 * d)
        sd[a + ".mha.out_proj.weight"] = rn(d, d); sd[a + ".mha.out_proj.bias"] = rn(d)
        sd[a + ".norm.weight"] = torch.ones(d) + rn(d); sd[a + ".norm.bias"] = rn(d)
        sd[ff + ".net.0.weight"] = rn(f, d); sd[ff + ".net.0.bias"] = rn(f)
        sd[ff + ".net.2.weight"] = rn(d, f); sd[ff + ".net.2.bias"] = rn(d)
        sd[ff + ".norm.weight"] = torch.ones(d) + rn(d); sd[ff + ".norm.bias"] = rn(d)
    sd["last_layer.0.weight"] = rn(d, d); sd["last_layer.0.bias"] = rn(d)
    sd["last_layer.2.weight"] = torch.ones(d) + rn(d); sd["last_layer.2.bias"] = rn(d)
    if cfg.kind == "bert":
        for q in range(cfg.splits):
            sd[f"bias.{q}"] = rn(cfg.seq, cfg.group_codes)
    else:
        sd["prediction_layer.weight"] = rn(cfg.splits * cfg.group_codes, d) * head_gain
        sd["prediction_layer.bias"] = rn(cfg.splits * cfg.group_codes)
    if cfg.prenorm:                   # drawn last: the streams of the post-norm configurations are unchanged
        sd["norm_after_transformer.weight"] = torch.ones(d) + rn(d); sd["norm_after_transformer.bias"] = rn(d)
    return sd


def decoder_plan(cfg: TokCfg) -> List[Tuple[str, int, int, bool]]:
    """(stage prefix, Cin, Cout, has_upsample) for decoder.up.* following autoencoder.py:370-392."""
    mult = tuple(cfg.channel_mult) + (cfg.channel_mult[-1],)
    out = []
    for s, lvl in enumerate(reversed(range(cfg.num_resolutions))):
        out.append((f"decoder.up.{s}", cfg.hidden_channels * mult[lvl + 1], cfg.hidden_channels * mult[lvl], lvl > 0))
    return out


def make_tokenizer_weights(cfg: TokCfg, seed: int = 0, with_encoder: bool = False) -> StateDict:
    """Seeded conv weights randn/sqrt(fan_in), GN gamma ~ 1, with the reference's key names."""
    g = torch.Generator().manual_seed(seed)

    def conv(co, ci, k):
        return torch.randn(co, ci, k, k, generator=g) / math.sqrt(ci * k * k)

    def vec(c, base=0.0, s=0.05):
        return base + torch.randn(c, generator=g) * s

    sd: StateDict = {}

    def res_block(p, ci, co):
        sd[p + ".norm1.weight"] = vec(ci, 1.0); sd[p + ".norm1.bias"] = vec(ci)
        sd[p + ".conv1.weight"] = conv(co, ci, 3)
        sd[p + ".norm2.weight"] = vec(co, 1.0); sd[p + ".norm2.bias"] = vec(co)
        sd[p + ".conv2.weight"] = conv(co, co, 3)
        if ci != co:
            sd[p + ".nin_shortcut.weight"] = conv(co, co, 1)

    top = cfg.hidden_channels * cfg.channel_mult[cfg.num_resolutions - 1]
    sd["decoder.conv_in.weight"] = conv(top, cfg.token_size, 3); sd["decoder.conv_in.bias"] = vec(top)
    for r in range(cfg.num_res_blocks):
        res_block(f"decoder.mid.res_blocks.{r}", top, top)
    last = top
    for p, ci, co, up in decoder_plan(cfg):
        c = ci
        for r in range(cfg.num_res_blocks):
            res_block(f"{p}.res_blocks.{r}", c, co)
            c = co
        if up:
            sd[p + ".upsample_conv.weight"] = conv(co, co, 3); sd[p + ".upsample_conv.bias"] = vec(co)
        last = co
    sd["decoder.norm_out.weight"] = vec(last, 1.0); sd["decoder.norm_out.bias"] = vec(last)
    sd["decoder.conv_out.weight"] = conv(cfg.num_channels, last, 3); sd["decoder.conv_out.bias"] = vec(cfg.num_channels, 0.5, 0.1)
    if with_encoder:
        emult = (1,) + tuple(cfg.channel_mult)
        sd["encoder.conv_in.weight"] = conv(cfg.hidden_channels, cfg.num_channels, 3)
        c = cfg.hidden_channels
        for s in range(cfg.num_resolutions):
            ci, co = cfg.hidden_channels * emult[s], cfg.hidden_channels * emult[s + 1]
            c = ci
            for r in range(cfg.num_res_blocks):
                res_block(f"encoder.down.{s}.res_blocks.{r}", c, co)
                c = co
            if s < cfg.num_resolutions - 1 and cfg.sample_with_conv:
                sd[f"encoder.down.{s}.down_conv.weight"] = conv(co, co, 3); sd[f"encoder.down.{s}.down_conv.bias"] = vec(co)
        for r in range(cfg.num_res_blocks):
            res_block(f"encoder.mid.res_blocks.{r}", c, c)
        sd["encoder.norm_out.weight"] = vec(c, 1.0); sd["encoder.norm_out.bias"] = vec(c)
        sd["encoder.conv_out.weight"] = conv(cfg.token_size, c, 1); sd["encoder.conv_out.bias"] = vec(cfg.token_size)
        sd["quantize.bits_to_indices"] = (1 << torch.arange(cfg.token_size)).to(torch.int32)
        sd["quantize.codebook"] = _index_to_bits(torch.arange(1 << cfg.token_size), cfg.token_size)
    return sd
