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
def make_generator_weights(cfg: GenCfg, seed: int = 0, head_gain: float = 1.0, style: str = "gaussian") -> StateDict:
    """Build-own seeded weights with the reference checkpoint's key names and shapes
    (SURVEY.md 8b).  randn*0.02 for Linear / Embedding / pos_emb, LN gamma=1 beta=0;
    ``head_gain`` scales prediction_layer.weight so the softmax is peaky enough to
    make logit errors visible in the sampled tokens.  ``style`` "outlier": see _trained_like."""
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
    if style != "gaussian":
        _trained_like(sd, cfg, seed, style)
    return sd


OUTLIER_CHANNELS = 6       # "trained-like" style: hidden channels that carry massive, nearly input-independent LayerNorm outputs
OUTLIER_BETA = 12.0        # ... of this magnitude (LayerNorm beta = +-OUTLIER_BETA, gamma x OUTLIER_GAMMA there): ~16x the rms of a normal channel
OUTLIER_GAMMA = 0.1
OUTLIER_CONSUMER = 0.0625  # the consumers' columns of those channels (an outlier channel then contributes about a normal channel's share of an output)


# 48 x 0.75, 12 x 1.5, 3 x 2.5, 1 x 4 of 64, normalised to unit second moment: kurtosis 10.9 (a Gaussian's: 3; Student-t_6: 6)
_HEAVY_TAIL_SCALES = torch.tensor([0.75] * 48 + [1.5] * 12 + [2.5] * 3 + [4.0]) / math.sqrt((48 * 0.75 ** 2 + 12 * 1.5 ** 2 + 3 * 2.5 ** 2 + 16.0) / 64)
# style "outlier2" (round 6: the held-out trained-like family): HEAVIER tails -- 48 x 0.7, 12 x 1.5, 3 x 3.0, 1 x 5.0 of 64: kurtosis 17.2, one weight in 64 is 7.1x
# the typical one -- and ten massive-activation channels instead of six
_HEAVIER_TAIL_SCALES = torch.tensor([0.7] * 48 + [1.5] * 12 + [3.0] * 3 + [5.0]) / math.sqrt((48 * 0.7 ** 2 + 12 * 1.5 ** 2 + 3 * 3.0 ** 2 + 25.0) / 64)
_STYLES = {"outlier": (_HEAVY_TAIL_SCALES, OUTLIER_CHANNELS), "outlier2": (_HEAVIER_TAIL_SCALES, 10)}


def _trained_like(sd: StateDict, cfg: GenCfg, seed: int, style: str) -> None:
    """Post-transform of the Gaussian draw towards the statistics trained transformers show and Gaussian weights do not (what per-row / per-block
    MX-fp4 scales and fp16 activations are sensitive to):  style "outlier" = (1) heavy-tailed Linear weights -- every 2-D trunk / head weight is
    multiplied elementwise by a random scale from _HEAVY_TAIL_SCALES (a scale mixture of normals with the Gaussian draw's standard deviation and
    kurtosis 10.9: one weight in 64 is 4.7x its draw, three are 2.9x);
    (2) OUTLIER_CHANNELS hidden channels carry massive LayerNorm outputs, the same channels in every layer ("massive activations"): beta =
    +-OUTLIER_BETA and gamma x OUTLIER_GAMMA in first_layer.0 and both norms of every layer -- in a post-norm trunk the residual stream then holds
    +-12 in those channels next to ~0.75 rms in the others, a stable fixed point at which the logits still depend on tokens and class as much as
    the Gaussian draw's do (a large gamma instead runs away -- the next LayerNorm divides everything else by the outliers' magnitude -- and from
    +-15 on the random trunk stops passing information) -- and the matching in_proj / net.0 / last_layer.0 columns x OUTLIER_CONSUMER.  Deterministic
    from the seed (a generator of its own, so the Gaussian draw and every existing fixture stay what they were)."""
    if style not in _STYLES:
        raise ValueError(f"unknown weight style '{style}'")
    tail_scales, n_outlier = _STYLES[style]
    g = torch.Generator().manual_seed(1_000_003 * (seed + 1))
    d = cfg.hidden
    for k in sorted(sd):
        v = sd[k]
        if v.dim() == 2 and (k.startswith("transformer.layers.") or k.startswith("last_layer.0") or k.startswith("prediction_layer")):
            # a scale mixture of normals from an integer draw and a table of exact constants: one multiply per weight, bit-reproducible on any host
            # (torch's CPU exponential_ and even sqrt are not: both differ between this build container and the GPU boxes' hosts)
            sd[k] = v * tail_scales[torch.randint(0, 64, v.shape, generator=g)]
    ch = torch.randperm(d, generator=g)[:n_outlier]
    sign = torch.where(torch.rand(n_outlier, generator=g) < 0.5, -1.0, 1.0)
    norms = ["first_layer.0"] + [f"transformer.layers.{l}.{s}.norm" for l in range(cfg.depth) for s in (0, 1)]
    for n in norms:
        sd[n + ".weight"][ch] *= OUTLIER_GAMMA
        sd[n + ".bias"][ch] = OUTLIER_BETA * sign
    for l in range(cfg.depth):
        sd[f"transformer.layers.{l}.0.mha.in_proj_weight"][:, ch] *= OUTLIER_CONSUMER
        sd[f"transformer.layers.{l}.1.net.0.weight"][:, ch] *= OUTLIER_CONSUMER
    sd["last_layer.0.weight"][:, ch] *= OUTLIER_CONSUMER       # the last layer's second norm feeds the head


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
