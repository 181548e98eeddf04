"""GPU parity of the tokenizer's encode half: ConvEncoder + lookup-free sign/pack, and
ConvVQModel.forward = decode(encode(x)).  The quantiser takes the SIGN of the encoder output, so a pre-activation within
the fp16 error of zero may land on the other side: bits are compared where the oracle's |z| is clear of that error, the
raw latent everywhere."""
import numpy as np
import pytest
import torch

from hip_helpers import hip_tokenizer
from oracle import maskbit_oracle as O
from test_oracle_golden import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda"
TINY_TOK = O.TokCfg(token_size=12, hidden_channels=64, channel_mult=(1, 1, 2), num_resolutions=3, num_res_blocks=1)


def _oracle_z(sd, cfg, x):
    """pre-sign encoder output of the oracle (same code as O.encode_image up to the quantiser)"""
    h = O._conv_same(x, sd["encoder.conv_in.weight"], None)
    for s in range(cfg.num_resolutions):
        for r in range(cfg.num_res_blocks):
            h = O._res_block(h, sd, f"encoder.down.{s}.res_blocks.{r}")
        if s < cfg.num_resolutions - 1:
            h = O._conv_same(h, sd[f"encoder.down.{s}.down_conv.weight"], sd[f"encoder.down.{s}.down_conv.bias"], stride=2)
    for r in range(cfg.num_res_blocks):
        h = O._res_block(h, sd, f"encoder.mid.res_blocks.{r}")
    h = O._gn_silu(h, sd, "encoder.norm_out")
    return O._conv_same(h, sd["encoder.conv_out.weight"], sd["encoder.conv_out.bias"])


def _check(tk, sd, cfg, x, ref_idx, tol_rel):
    zq, idx, zraw = tk._encode(x.to(DEV), want_raw=True)
    zref = _oracle_z(sd, cfg, x)
    scale = float(zref.abs().mean())
    err = float((zraw.cpu() - zref).abs().max())
    print(f"encoder latent: max err {err:.3e} (mean |z| {scale:.3f})")
    assert err < tol_rel * scale
    K = cfg.token_size
    bits = (idx.cpu()[..., None] >> torch.arange(K)) & 1                       # [b,h,w,K]
    ref_bits = (ref_idx[..., None] >> torch.arange(K)) & 1
    clear = (zref.permute(0, 2, 3, 1).abs() > 2 * err)
    assert bool((bits == ref_bits)[clear].all())                               # every decidable bit agrees with the reference
    assert float((bits != ref_bits).float().mean()) < 0.02
    assert torch.equal(zq.cpu(), torch.where(zraw.cpu() > 0, 1.0, -1.0))       # lookup_free.py:57-59
    w = (2 ** torch.arange(K)).view(1, K, 1, 1)
    assert torch.equal(((zq.cpu() > 0).long() * w).sum(1), idx.cpu())          # lookup_free.py:113-127, LSB first
    return idx


def test_encoder_tiny_vs_reference_golden():
    z = load_golden("tok_tiny.npz")
    sd = O.make_tokenizer_weights(TINY_TOK, seed=int(z["seed"]), with_encoder=True)
    tk = hip_tokenizer(TINY_TOK, sd)
    x = torch.from_numpy(z["enc_input"])
    _check(tk, sd, TINY_TOK, x, torch.from_numpy(z["enc_indices"]).long(), 0.03)


def test_encoder_config1_10bit_full_size_and_forward():
    """BASELINE configs[0] on the GPU: 10-bit tokenizer, encode + decode one 256x256 image."""
    z = load_golden("tok_full10_cfg1.npz")
    cfg = O.TokCfg(token_size=10)
    sd = O.make_tokenizer_weights(cfg, seed=int(z["seed"]), with_encoder=True)
    tk = hip_tokenizer(cfg, sd)
    x = torch.rand(1, 3, 256, 256, generator=torch.Generator().manual_seed(0))
    idx = _check(tk, sd, cfg, x, torch.from_numpy(z["indices"]).long(), 0.03)
    # API surface: encode -> (z_quantized, result_dict), forward -> (reconstruction, result_dict)
    zq, res = tk.encode(x.to(DEV))
    assert zq.shape == (1, 10, 16, 16) and torch.equal(res["min_encoding_indices"], idx)
    assert set(res) == {"quantizer_loss", "commitment_loss", "entropy_loss", "per_sample_entropy", "avg_entropy", "min_encoding_indices"}
    rec, res2 = tk(x.to(DEV))
    assert rec.shape == (1, 3, 256, 256) and torch.equal(res2["min_encoding_indices"], idx)
    want = O.decode_tokens(sd, cfg, idx.cpu().reshape(1, -1))
    assert float((rec.cpu() - want).abs().max()) < 0.03
    # batch > 1 and determinism
    xb = torch.cat([x, torch.rand(2, 3, 256, 256, generator=torch.Generator().manual_seed(5))]).to(DEV)
    _, rb = tk.encode(xb)
    assert torch.equal(rb["min_encoding_indices"][:1], idx)
    _, rb2 = tk.encode(xb)
    assert torch.equal(rb["min_encoding_indices"], rb2["min_encoding_indices"])


def test_encoder_rejects_bad_input():
    z = load_golden("tok_tiny.npz")
    sd = O.make_tokenizer_weights(TINY_TOK, seed=int(z["seed"]), with_encoder=True)
    tk = hip_tokenizer(TINY_TOK, sd)
    with pytest.raises(ValueError):
        tk.encode(torch.zeros(1, 3, 60, 64, device=DEV))
    with pytest.raises(ValueError):
        tk.encode(torch.zeros(1, 4, 64, 64, device=DEV))


def test_encoder_average_pool_variant_vs_reference_golden():
    """sample_with_conv=False (autoencoder.py:179-182): avg_pool2d(2, 2) between the encoder stages; golden from the reference."""
    from oracle.make_golden_variants import AVGPOOL_TOK as cfg
    z = load_golden("tok_avgpool_tiny.npz")
    sd = O.make_tokenizer_weights(cfg, seed=int(z["seed"]), with_encoder=True)
    tk = hip_tokenizer(cfg, sd)
    x = torch.from_numpy(z["enc_input"])
    zq, idx, zraw = tk._encode(x.to(DEV), want_raw=True)
    # pre-sign latent against the oracle (same stages, pooling instead of the strided conv)
    h = O._conv_same(x, sd["encoder.conv_in.weight"], None)
    for s in range(cfg.num_resolutions):
        for r in range(cfg.num_res_blocks):
            h = O._res_block(h, sd, f"encoder.down.{s}.res_blocks.{r}")
        if s < cfg.num_resolutions - 1:
            h = torch.nn.functional.avg_pool2d(h, 2, 2)
    for r in range(cfg.num_res_blocks):
        h = O._res_block(h, sd, f"encoder.mid.res_blocks.{r}")
    zref = O._conv_same(O._gn_silu(h, sd, "encoder.norm_out"), sd["encoder.conv_out.weight"], sd["encoder.conv_out.bias"])
    err = float((zraw.cpu() - zref).abs().max())
    assert err < 0.03 * float(zref.abs().mean())
    K = cfg.token_size
    bits = (idx.cpu()[..., None] >> torch.arange(K)) & 1
    ref_bits = (torch.from_numpy(z["enc_indices"]).long()[..., None] >> torch.arange(K)) & 1
    clear = zref.permute(0, 2, 3, 1).abs() > 2 * err
    assert bool((bits == ref_bits)[clear].all()) and float((bits != ref_bits).float().mean()) < 0.02
    # decode of the reference's own codes against the reference's reconstruction
    rec = tk.decode(torch.from_numpy(z["enc_zq"]).float().to(DEV))
    assert float((rec.cpu() - torch.from_numpy(z["recon"]).float()).abs().max()) < 0.03
