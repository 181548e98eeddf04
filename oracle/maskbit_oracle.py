"""CPU oracle for the MaskBit sampling hot path.  TEST INFRASTRUCTURE ONLY.

This module is a plain PyTorch-CPU fp32 restatement of the reference algorithm
(markweberdev/maskbit) for the path named in BASELINE.json: bit-token embed ->
24-layer bidirectional transformer -> per-bit-group logits -> categorical /
Gumbel-confidence sampling with cosine-schedule re-mask -> conv-VQGAN decode.
It exists so that the HIP path can be checked against *something that runs on
the GPU box* (the reference itself cannot travel).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it; the product package ``maskbit_amd`` never does.

Parity pin: every function here is checked against golden vectors captured by
importing the real reference in the build container (``oracle/make_golden.py``
-> ``tests/golden/*.npz``; see ``tests/test_oracle_golden.py``).  The reference
ships no tests of its own for this path apart from two ``__main__`` blocks
(factorization.py:49-67, lookup_free.py:146-163), which are restated as tests.

It is written functionally over a flat ``state_dict`` (name -> tensor) using the
reference's checkpoint key names, so the same weights feed the oracle and the
HIP engine.  Every function cites the reference lines it follows
(paths relative to the reference root).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
StateDict = Dict[str, Tensor]


# --------------------------------------------------------------------------- configs + seeded synthetic weights
# (shared with bench.py and the HIP tests: maskbit_amd/synth.py holds the config dataclasses and the weight generators; no algorithm there)
from maskbit_amd.synth import GenCfg, TokCfg, decoder_plan, make_generator_weights, make_tokenizer_weights  # noqa: E402,F401


# --------------------------------------------------------------------------- bit helpers
def index_to_bits(idx: Tensor, nbits: int) -> Tensor:
    """LSB-first {-1,+1} expansion: lookup_free.py:96-111 / bert.py:449-450."""
    weights = (1 << torch.arange(nbits, dtype=torch.int64, device=idx.device))
    on = (idx.long().unsqueeze(-1) & weights) != 0
    return on.to(torch.float32) * 2.0 - 1.0


def bits_to_index(bits: Tensor) -> Tensor:
    """sign -> integer code, LSB-first (lookup_free.py:113-127)."""
    nbits = bits.shape[-1]
    weights = (1 << torch.arange(nbits, dtype=torch.int64, device=bits.device))
    return ((bits > 0).long() * weights).sum(-1)


def combine_groups(tokens: Tensor, bits: int, splits: int) -> Tensor:
    """m group indices -> one K-bit index, returned as float32 (factorization.py:7-24)."""
    shift = bits // splits
    out = torch.zeros(tokens.shape[:2], dtype=torch.float32, device=tokens.device)
    for g in range(splits):
        out += (tokens[..., g] << (g * shift))
    return out


def split_groups(tokens: Tensor, bits: int, splits: int) -> Tensor:
    """inverse of combine_groups (factorization.py:27-46)."""
    shift = bits // splits
    low = (1 << shift) - 1
    return torch.stack([(tokens >> (g * shift)) & low for g in range(splits)], dim=2)


# --------------------------------------------------------------------------- generator
def _ln(x: Tensor, sd: StateDict, prefix: str, eps: float) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[prefix + ".weight"], sd[prefix + ".bias"], eps)


def token_bit_vectors(tokens: Tensor, cfg: GenCfg) -> Tensor:
    """bert.py:440-454: [b,n,m] int -> [b,n,K] in {-1,0,+1}; channel = g*(K/m)+j."""
    gb = cfg.group_bits
    v = index_to_bits(tokens, gb)                                  # [b,n,m,gb]
    v = torch.where((tokens == cfg.group_codes).unsqueeze(-1), torch.zeros_like(v), v)
    return v.reshape(tokens.shape[0], tokens.shape[1], cfg.splits * gb)


def attention(x: Tensor, sd: StateDict, p: str, heads: int, attn_out: Optional[list] = None) -> Tensor:
    """nn.MultiheadAttention(batch_first, packed in_proj) self-attention (bert.py:84,137).  With ``attn_out`` the attention
    weights averaged over the heads, [b, n, n], are appended to it (``need_weights=True`` with the module's default
    ``average_attn_weights=True``, bert.py:119,137)."""
    b, n, d = x.shape
    dh = d // heads
    qkv = F.linear(x, sd[p + ".in_proj_weight"], sd[p + ".in_proj_bias"])
    q, k, v = qkv.split(d, dim=-1)
    q = q.reshape(b, n, heads, dh).transpose(1, 2)
    k = k.reshape(b, n, heads, dh).transpose(1, 2)
    v = v.reshape(b, n, heads, dh).transpose(1, 2)
    s = (q * (1.0 / math.sqrt(dh))) @ k.transpose(-1, -2)
    w = torch.softmax(s, dim=-1)
    if attn_out is not None:
        attn_out.append(w.mean(dim=1))
    o = w @ v
    o = o.transpose(1, 2).reshape(b, n, d)
    return F.linear(o, sd[p + ".out_proj.weight"], sd[p + ".out_proj.bias"])


def _trunk(sd: StateDict, cfg: GenCfg, x: Tensor, attn_out: Optional[list] = None) -> Tensor:
    """first_layer LayerNorm, TransformerEncoder (post- or pre-norm), norm_after_transformer (pre-norm only), last_layer
    (bert.py:27-70, 84-141, 166-180, 496-500)."""
    x = _ln(x, sd, "first_layer.0", 1e-12)
    for l in range(cfg.depth):
        a = f"transformer.layers.{l}.0"
        f = f"transformer.layers.{l}.1"
        if cfg.prenorm:                                                                    # :49-59, :106-123
            x = attention(_ln(x, sd, a + ".norm", 1e-12), sd, a + ".mha", cfg.heads, attn_out) + x
            y = _ln(x, sd, f + ".norm", 1e-12)
            x = F.linear(F.gelu(F.linear(y, sd[f + ".net.0.weight"], sd[f + ".net.0.bias"])), sd[f + ".net.2.weight"], sd[f + ".net.2.bias"]) + x
        else:
            x = _ln(attention(x, sd, a + ".mha", cfg.heads, attn_out) + x, sd, a + ".norm", 1e-12)      # :137-139
            h = F.gelu(F.linear(x, sd[f + ".net.0.weight"], sd[f + ".net.0.bias"]))           # erf GELU
            x = _ln(F.linear(h, sd[f + ".net.2.weight"], sd[f + ".net.2.bias"]) + x, sd, f + ".norm", 1e-12)
    if cfg.prenorm:
        x = _ln(x, sd, "norm_after_transformer", 1e-12)                                    # :498-499
    x = F.gelu(F.linear(x, sd["last_layer.0.weight"], sd["last_layer.0.bias"]))
    return _ln(x, sd, "last_layer.2", 1e-12)


def lfq_bert_forward(sd: StateDict, cfg: GenCfg, tokens: Tensor, labels: Tensor,
                     drop: Optional[Tensor] = None, return_attn: bool = False):
    """LFQBert.forward (bert.py:456-508) or, for cfg.kind == "bert", Bert.forward (bert.py:283-340). Returns [b,seq,m,C] fp32;
    with ``return_attn`` (bert.py:505-508 / 337-340) a tuple (logits, [per-layer head-averaged attention [b, seq+1, seq+1]])."""
    if cfg.kind == "bert":
        return bert_forward(sd, cfg, tokens, labels, drop, return_attn)
    b = tokens.shape[0]
    lab = labels.long().clone()
    if drop is not None:
        lab = torch.where(drop.bool(), torch.full_like(lab, cfg.nclass), lab)     # :482-484
    x_tok = F.linear(token_bit_vectors(tokens, cfg), sd["input_proj.weight"], sd["input_proj.bias"])
    x_cls = sd["class_emb.weight"][lab].unsqueeze(1)
    x = torch.cat([x_tok, x_cls], dim=1) + sd["pos_emb"]                          # class row LAST
    attn = [] if return_attn else None
    x = _trunk(sd, cfg, x, attn)
    logits = F.linear(x, sd["prediction_layer.weight"], sd["prediction_layer.bias"])
    logits = logits.reshape(b, cfg.seq + 1, cfg.splits, cfg.group_codes)
    return (logits[:, :cfg.seq], attn) if return_attn else logits[:, :cfg.seq]


def bert_forward(sd: StateDict, cfg: GenCfg, tokens: Tensor, labels: Tensor, drop: Optional[Tensor] = None, return_attn: bool = False):
    """Bert.forward (bert.py:283-340): per-group embedding tables summed, output head tied to them plus a per-position bias."""
    lab = labels.long().clone()
    if drop is not None:
        lab = torch.where(drop.bool(), torch.full_like(lab, cfg.nclass), lab)     # :309-311
    x_tok = sd["tok_emb_list.0.weight"][tokens[..., 0]]
    for g in range(1, cfg.splits):
        x_tok = x_tok + sd[f"tok_emb_list.{g}.weight"][tokens[..., g]]            # :313-315
    x_cls = sd["class_emb.weight"][lab].unsqueeze(1)
    x = torch.cat([x_tok, x_cls], dim=1) + sd["pos_emb"]
    attn = [] if return_attn else None
    x = _trunk(sd, cfg, x, attn)
    C_ = cfg.group_codes
    logits = [torch.matmul(x, sd[f"tok_emb_list.{g}.weight"].t()[:, :C_])[:, :cfg.seq] + sd[f"bias.{g}"] for g in range(cfg.splits)]   # :329-332
    out = torch.stack(logits, dim=2)
    return (out, attn) if return_attn else out


# --------------------------------------------------------------------------- schedule
def masking_ratio(progress: float, mode: str = "arccos") -> Tensor:
    """get_masking_ratio (masking.py:41-65): float32 torch scalar, clamp [1e-6, 1]."""
    r = torch.tensor(progress)
    if mode == "root":
        v = 1 - (r ** 0.5)
    elif mode == "square":
        v = 1 - (r ** 2)
    elif mode == "cosine":
        v = torch.cos(r * math.pi * 0.5)
    elif mode == "arccos":
        v = torch.acos(r) / (math.pi * 0.5)
    elif mode == "linear":
        v = 1 - r
    else:
        raise ValueError("Invalid mode. Choose between 'linear','square', 'cosine', 'arccos', 'root'.")
    return torch.clamp(v, 1e-6, 1.0)


def guidance_factor(i: int, num_steps: int, annealing: str, scale_pow: float) -> Tensor:
    """Per-step CFG multiplier a_i (sampling.py:91-97); uses i/N, not progress."""
    if annealing == "none":
        return torch.tensor(1.0)
    if annealing == "linear":
        return torch.tensor(i / num_steps)
    if annealing == "cosine":
        sp = torch.ones(1) * scale_pow
        return (1 - torch.cos(((i / num_steps) ** sp) * torch.pi)) * 1 / 2
    raise ValueError(f"unknown guidance_annealing {annealing!r}")


def mask_len_schedule(num_steps: int, num_maskable: int, mode: str = "arccos") -> List[float]:
    """floor(ratio*num_maskable) per step (sampling.py:120-123), before the [1, num_masked-1] clamp."""
    return [float(torch.floor(masking_ratio((i + 1) / num_steps, mode) * num_maskable)) for i in range(num_steps)]


# --------------------------------------------------------------------------- one sampling step
def sample_step(logits_c: Tensor, logits_u: Optional[Tensor], scale, softmax_temperature: float,
                exp_noise: Tensor, conf_noise: Tensor, tokens: Tensor, mask_token: int,
                mask_ratio: Tensor, num_maskable: int) -> Tuple[Tensor, Tensor]:
    """One iteration body of sample() after the forward (sampling.py:98-131).

    exp_noise  [B*n*m, C]: the Exp(1) draw torch.multinomial(n=1) would make (argmax(p/q)).
    conf_noise [B,n,m]   : gumbel * randomize_temperature * (1-progress), already scaled.
    Returns (pred [B,n,m] int64, new masked tokens [B,n,m] int64).
    """
    B = tokens.shape[0]
    if logits_u is not None:
        logits = logits_c + scale * (logits_c - logits_u)                      # :98-99
    else:
        logits = logits_c
    p = torch.softmax(logits / softmax_temperature, dim=-1)                     # :105
    pn = p / p.sum(-1, keepdim=True)                                            # Categorical.__init__
    pred = torch.argmax(pn.reshape(-1, p.shape[-1]) / exp_noise, dim=-1).reshape(tokens.shape)   # multinomial(n=1)
    mask = tokens == mask_token
    num_masked = mask.sum(dim=(1, 2))[0]                                        # sample 0 only, :109
    pred = torch.where(mask, pred, tokens)                                      # :111
    conf = torch.gather(p, -1, pred.unsqueeze(-1)).squeeze(-1)                  # :113
    conf = torch.where(mask, conf, torch.inf)
    conf = torch.log(conf) + conf_noise                                         # :117-118
    mask_len = torch.floor(mask_ratio * num_maskable)
    k = torch.clamp(mask_len, torch.ones_like(num_masked), num_masked - 1).long()   # :123-124
    thr = torch.sort(conf.view(B, -1), dim=-1).values[:, k - 1]
    new_tokens = torch.where(conf <= thr.view(B, 1, 1), mask_token, pred)       # :128-129
    return pred, new_tokens


@dataclass
class StepRecord:
    logits_c: Tensor
    logits_u: Optional[Tensor]
    scale: float
    exp_noise: Tensor
    conf_noise: Tensor
    tokens_in: Tensor
    mask_ratio: float
    pred: Tensor
    tokens_out: Tensor


def sample_loop(forward, num_samples: int, labels: Tensor, *, softmax_temperature: float = 1.0,
                randomize_temperature: float = 4.5, mask_schedule_strategy: str = "linear",
                num_steps: int = 12, guidance_scale: float = 3.0, mask_token: int = 1024,
                patch_size: int = 16, guidance_annealing: str = "none",
                use_sampling_annealing: bool = False, scale_pow: float = 4.0,
                codebook_splits: int = 1, record: Optional[List[StepRecord]] = None,
                noise: Optional[Sequence[Tuple[Tensor, Tensor]]] = None) -> List[Tensor]:
    """The N-step loop of sample() (sampling.py:55-131) on CPU, RNG protocol included.

    ``forward(tokens, labels, drop) -> logits``.  When ``noise`` is None the draws follow
    the reference's order on a CPU model: per step (1) exponential_ of shape [B*n*m, C] from
    the default generator (inside torch.multinomial), (2) Gumbel(0,1).sample([B,n,m]).
    When ``noise`` is given it is a per-step list of (exp_noise, gumbel) and nothing is drawn.
    Returns the per-step ``pred`` tokens (the reference's l_full_tokens).
    """
    n = int(patch_size ** 2)
    m = int(codebook_splits)
    drop = torch.ones(num_samples, dtype=torch.bool)
    tokens = torch.full((num_samples, n, m), mask_token, dtype=torch.int64)
    num_maskable = n * m
    gumbel = torch.distributions.Gumbel(loc=0.0, scale=1.0)
    preds: List[Tensor] = []
    for i in range(num_steps):
        progress = (i + 1) / num_steps
        if guidance_scale != 0.0:
            lg = forward(torch.cat([tokens, tokens]), torch.cat([labels, labels]), torch.cat([~drop, drop]))
            lc, lu = torch.chunk(lg, 2, dim=0)
            scale = guidance_scale * guidance_factor(i, num_steps, guidance_annealing, scale_pow)
        else:
            lc, lu, scale = forward(tokens, labels, ~drop), None, torch.tensor(0.0)
        if use_sampling_annealing:
            softmax_temperature = 0.5 + 0.8 * (1 - progress)
        C = lc.shape[-1]
        if noise is None:
            q = torch.empty(num_samples * n * m, C).exponential_(1)
            g = gumbel.sample((num_samples, n, m))
        else:
            q, g = noise[i]
        cn = g * randomize_temperature * (1 - progress)
        ratio = masking_ratio(progress, mask_schedule_strategy)
        pred, new_tokens = sample_step(lc, lu, scale, softmax_temperature, q, cn, tokens, mask_token,
                                       ratio, num_maskable)
        if record is not None:
            record.append(StepRecord(lc, lu, float(scale), q, cn, tokens, float(ratio), pred, new_tokens))
        tokens = new_tokens
        preds.append(pred)
    return preds


# --------------------------------------------------------------------------- tokenizer (decoder + encoder)
def _conv_same(x: Tensor, w: Tensor, b: Optional[Tensor], stride: int = 1) -> Tensor:
    """Conv2dSame (autoencoder.py:7-36): TF-style SAME padding, extra pixel on the bottom/right."""
    k = w.shape[-1]
    ih, iw = x.shape[-2:]
    ph = max((math.ceil(ih / stride) - 1) * stride + (k - 1) + 1 - ih, 0)
    pw = max((math.ceil(iw / stride) - 1) * stride + (k - 1) + 1 - iw, 0)
    if ph or pw:
        x = F.pad(x, [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2])
    return F.conv2d(x, w, b, stride=stride)


def _gn_silu(x: Tensor, sd: StateDict, p: str) -> Tensor:
    return F.silu(F.group_norm(x, 32, sd[p + ".weight"], sd[p + ".bias"], 1e-6))     # autoencoder.py:39-43


def _res_block(x: Tensor, sd: StateDict, p: str) -> Tensor:
    """ResidualBlock (autoencoder.py:76-96). NB the shortcut quirk: when Cin != Cout the 1x1
    conv is applied to the block OUTPUT h and the input is dropped: out = h + W*h."""
    h = _conv_same(_gn_silu(x, sd, p + ".norm1"), sd[p + ".conv1.weight"], None)
    h = _conv_same(_gn_silu(h, sd, p + ".norm2"), sd[p + ".conv2.weight"], None)
    if (p + ".nin_shortcut.weight") in sd:
        return h + _conv_same(h, sd[p + ".nin_shortcut.weight"], None)
    return h + x


def decode_latents(sd: StateDict, cfg: TokCfg, z: Tensor) -> Tensor:
    """ConvDecoder.forward (autoencoder.py:399-423): z [b,K,h,w] -> image [b,3,H,W]."""
    nrb = cfg.num_res_blocks
    x = _conv_same(z, sd["decoder.conv_in.weight"], sd["decoder.conv_in.bias"])
    for r in range(nrb):
        x = _res_block(x, sd, f"decoder.mid.res_blocks.{r}")
    for s in range(cfg.num_resolutions):               # up.0 is the coarsest level (i_level = R-1)
        for r in range(nrb):
            x = _res_block(x, sd, f"decoder.up.{s}.res_blocks.{r}")
        if s < cfg.num_resolutions - 1:                # UpsamplingStage: nearest x2 then conv (:224-225)
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
            x = _conv_same(x, sd[f"decoder.up.{s}.upsample_conv.weight"], sd[f"decoder.up.{s}.upsample_conv.bias"])
    x = _gn_silu(x, sd, "decoder.norm_out")
    return _conv_same(x, sd["decoder.conv_out.weight"], sd["decoder.conv_out.bias"])


def decode_tokens(sd: StateDict, cfg: TokCfg, tokens: Tensor) -> Tensor:
    """ConvVQModel.decode_tokens (conv_vqgan.py:98-112): tokens [b,n] (any numeric dtype)."""
    z = index_to_bits(tokens.long(), cfg.token_size)              # [b,n,K]
    side = int(math.sqrt(float(z.shape[1])))
    z = z.reshape(z.shape[0], side, side, -1).permute(0, 3, 1, 2).contiguous()
    return decode_latents(sd, cfg, z)


def encode_image(sd: StateDict, cfg: TokCfg, x: Tensor) -> Tuple[Tensor, Tensor]:
    """ConvEncoder.forward + LFQ sign/pack (autoencoder.py:274-286, lookup_free.py:57-62).
    Returns (z_quantized [b,K,h,w] in {-1,+1}, indices [b,h,w]).  BASELINE config 1 plumbing."""
    nrb = cfg.num_res_blocks
    h = _conv_same(x, sd["encoder.conv_in.weight"], None)
    for s in range(cfg.num_resolutions):
        for r in range(nrb):
            h = _res_block(h, sd, f"encoder.down.{s}.res_blocks.{r}")
        if s < cfg.num_resolutions - 1:
            if cfg.sample_with_conv:
                h = _conv_same(h, sd[f"encoder.down.{s}.down_conv.weight"], sd[f"encoder.down.{s}.down_conv.bias"], stride=2)
            else:
                h = F.avg_pool2d(h, kernel_size=2, stride=2)
    for r in range(nrb):
        h = _res_block(h, sd, f"encoder.mid.res_blocks.{r}")
    h = _gn_silu(h, sd, "encoder.norm_out")
    z = _conv_same(h, sd["encoder.conv_out.weight"], sd["encoder.conv_out.bias"])
    zq = torch.where(z > 0.0, torch.ones_like(z), -torch.ones_like(z))
    idx = bits_to_index(zq.permute(0, 2, 3, 1))
    return zq, idx


def to_uint8_nhwc(img: Tensor) -> Tensor:
    """Caller post-processing (eval_maskbit.py:134-135): clamp, *255, NHWC, truncating cast."""
    return (torch.clamp(img, 0.0, 1.0) * 255.0).permute(0, 2, 3, 1).to(torch.uint8)


# --------------------------------------------------------------------------- whole path
def sample(gen_sd: StateDict, gen_cfg: GenCfg, tok_sd: StateDict, tok_cfg: TokCfg, num_samples: int,
           labels: Tensor, **kw) -> Tuple[Tensor, List[Tensor]]:
    """modeling.modules.sample (sampling.py:13-136) on the oracle models."""
    kw.setdefault("mask_token", gen_cfg.group_codes)
    kw.setdefault("codebook_splits", gen_cfg.splits)
    fwd = lambda t, y, d: lfq_bert_forward(gen_sd, gen_cfg, t, y, d)
    preds = sample_loop(fwd, num_samples, labels, **kw)
    combined = combine_groups(preds[-1], gen_cfg.bits, gen_cfg.splits)
    return decode_tokens(tok_sd, tok_cfg, combined), preds
