"""Pin the CPU oracle against golden vectors captured from the real reference
(oracle/make_golden.py).  CPU only; runs everywhere."""
import hashlib

import numpy as np
import pytest
import torch

from conftest import load_golden, golden_weights
from oracle import maskbit_oracle as O

TINY_GEN = O.GenCfg(bits=12, splits=2, hidden=128, depth=2, heads=4, mlp=256, seq=256, nclass=10)
TINY_TOK = O.TokCfg(token_size=12, hidden_channels=64, channel_mult=(1, 1, 2), num_resolutions=3, num_res_blocks=1)


def tiny_tok_weights():
    z = load_golden("tok_tiny.npz")
    sd = O.make_tokenizer_weights(TINY_TOK, seed=int(z["seed"]), with_encoder=True)
    assert sha(sd["decoder.conv_in.weight"]) == str(z["w_sha_conv_in"])
    assert sha(sd["encoder.conv_in.weight"]) == str(z["w_sha_enc_conv_in"])
    return sd


def sha(t):
    return hashlib.sha256(t.detach().contiguous().numpy().tobytes()).hexdigest()


def test_rng_sentinels():
    z = load_golden("rng_sentinel.npz")
    torch.manual_seed(1234)
    a = torch.empty(1536, 64).exponential_(1)
    b = torch.rand(3, 256, 2)
    assert sha(a) == str(z["exp_sha"]) and sha(b) == str(z["rand_sha"])
    assert sha(a)[:16] == "7780fa91e0fb77bc" and sha(b)[:16] == "9d43bf6dc222258c"   # SURVEY 8c probe


def test_generator_forward_tiny():
    z = load_golden("gen_tiny.npz")
    sd = golden_weights(z)
    regen = O.make_generator_weights(TINY_GEN, seed=11, head_gain=40.0)
    assert set(regen) == set(sd) and all(torch.equal(regen[k], sd[k]) for k in sd)
    out = O.lfq_bert_forward(sd, TINY_GEN, torch.from_numpy(z["tokens"]), torch.from_numpy(z["labels"]),
                             torch.from_numpy(z["drop"]))
    ref = torch.from_numpy(z["logits"])
    assert out.shape == ref.shape
    assert (out - ref).abs().max().item() < 2e-4          # fp32 op-order noise at |logit| ~ 40


def test_generator_does_not_mutate_labels():
    z = load_golden("gen_tiny.npz")
    sd = golden_weights(z)
    lab = torch.from_numpy(z["labels"]).clone()
    O.lfq_bert_forward(sd, TINY_GEN, torch.from_numpy(z["tokens"]), lab, torch.from_numpy(z["drop"]))
    assert torch.equal(lab, torch.from_numpy(z["labels"]))


def test_decode_and_encode_tiny():
    z = load_golden("tok_tiny.npz")
    sd = tiny_tok_weights()
    img = O.decode_tokens(sd, TINY_TOK, torch.from_numpy(z["tokens"]).float())
    assert (img - torch.from_numpy(z["image"])).abs().max().item() < 1e-4
    zq, idx = O.encode_image(sd, TINY_TOK, torch.from_numpy(z["enc_input"]))
    assert torch.equal(idx, torch.from_numpy(z["enc_indices"]).long())
    assert torch.equal(zq, torch.from_numpy(z["enc_zq"]))
    rec = O.decode_latents(sd, TINY_TOK, zq)
    assert (rec - torch.from_numpy(z["recon"])).abs().max().item() < 1e-4


@pytest.mark.parametrize("name", ["sample_tiny_cfg", "sample_tiny_nocfg", "sample_tiny_linear_anneal", "sample_tiny_none_cfg"])   # (the last: the demo's call site, demo_utils.py:139-157)
def test_sample_loop_bit_exact(name):
    """Same seed, same RNG draw order -> the per-step tokens of the reference, bit for bit."""
    z = load_golden(name + ".npz")
    gsd = golden_weights(load_golden("gen_tiny.npz"))
    tsd = tiny_tok_weights()
    kw = {}
    for k, v in zip(z["kw_keys"], z["kw_vals"]):
        v = str(v)
        kw[str(k)] = (v == "True") if v in ("True", "False") else (int(v) if v.isdigit() else (float(v) if v.replace(".", "").isdigit() else v))
    torch.manual_seed(int(z["seed"]))
    img, steps = O.sample(gsd, TINY_GEN, tsd, TINY_TOK, 3, torch.from_numpy(z["labels"]), softmax_temperature=1.0,
                          patch_size=16, **kw)
    ref_steps = torch.from_numpy(z["steps"])
    assert len(steps) == ref_steps.shape[0]
    for i, s in enumerate(steps):
        assert torch.equal(s, ref_steps[i]), f"step {i}: {(s != ref_steps[i]).sum().item()} tokens differ"
    assert (img - torch.from_numpy(z["image"])).abs().max().item() < 1e-4
    u8 = O.to_uint8_nhwc(img)
    assert (u8.int() - torch.from_numpy(z["image_u8"]).int()).abs().max().item() <= 1     # truncating cast near .0


def test_schedule_tables():
    z = load_golden("schedule.npz")
    for mode in ("arccos", "cosine", "linear", "square", "root"):
        for N in (16, 64, 128, 256):
            assert O.mask_len_schedule(N, 512, mode) == list(z[f"{mode}_{N}"])
            mine = np.array([float(O.masking_ratio((i + 1) / N, mode)) for i in range(N)], dtype=np.float32)
            assert np.array_equal(mine, z[f"ratio_{mode}_{N}"])
    ks = O.mask_len_schedule(64, 512, "arccos")
    assert ks[:3] == [506.0, 501.0, 496.0] and ks[-4:] == [100.0, 81.0, 57.0, 0.0]    # SURVEY A11 (last clamps to 1)
    with pytest.raises(ValueError):
        O.masking_ratio(0.5, "bogus")


def test_factorization_known_answers():
    """Restates the reference's own __main__ checks (factorization.py:49-67, lookup_free.py:146-163)."""
    z = load_golden("schedule.npz")
    t = torch.from_numpy(z["split_in"])
    sp = O.split_groups(t, 12, 2)
    assert torch.equal(sp, torch.from_numpy(z["split_out"]))
    comb = O.combine_groups(sp, 12, 2)
    assert comb.dtype == torch.float32 and torch.equal(comb, torch.from_numpy(z["combine_out"]))
    assert torch.equal(comb.long(), t)
    t10 = torch.randint(0, 1023, (1, 16))
    s10 = O.split_groups(t10, 10, 2)
    assert torch.equal(t10 >> 5, s10[..., 1]) and torch.equal(t10 & 31, s10[..., 0])
    allc = torch.arange(1024)
    assert torch.equal(O.bits_to_index(O.index_to_bits(allc, 10)), allc)
    assert O.index_to_bits(torch.tensor([1]), 10)[0, 0] == 1.0 and O.index_to_bits(torch.tensor([1]), 10)[0, 1] == -1.0


def test_truncating_uint8():
    img = torch.tensor([0.999, 0.5, -0.2, 1.3]).view(1, 1, 2, 2).expand(1, 3, 2, 2)
    u8 = O.to_uint8_nhwc(img)
    assert u8[0, 0, 0, 0] == 254 and u8[0, 0, 1, 0] == 127 and u8[0, 1, 0, 0] == 0 and u8[0, 1, 1, 0] == 255


@pytest.mark.timeout(600)
def test_generator_forward_full12():
    z = load_golden("gen_full12.npz")
    cfg = O.GenCfg(bits=12, splits=2)
    sd = O.make_generator_weights(cfg, seed=int(z["seed"]), head_gain=float(z["head_gain"]))
    assert sha(sd["transformer.layers.0.0.mha.in_proj_weight"]) == str(z["w_sha_in_proj0"])
    assert sha(sd["prediction_layer.weight"]) == str(z["w_sha_pred"])
    out = O.lfq_bert_forward(sd, cfg, torch.from_numpy(z["tokens"]), torch.from_numpy(z["labels"]), torch.from_numpy(z["drop"]))
    assert (out - torch.from_numpy(z["logits"])).abs().max().item() < 5e-4


@pytest.mark.timeout(600)
def test_decode_full12_and_config1():
    z = load_golden("tok_full12.npz")
    cfg = O.TokCfg(token_size=12)
    sd = O.make_tokenizer_weights(cfg, seed=int(z["seed"]))
    assert sha(sd["decoder.conv_in.weight"]) == str(z["w_sha_conv_in"])
    img = O.decode_tokens(sd, cfg, torch.from_numpy(z["tokens"]))
    for (y, x) in ((0, 0), (120, 120), (240, 240), (37, 201)):
        assert (img[:, :, y:y + 16, x:x + 16] - torch.from_numpy(z[f"crop_{y}_{x}"])).abs().max().item() < 2e-4
    assert np.allclose(img.mean((0, 2, 3)).numpy(), z["mean"], atol=1e-5)
    assert (img[:, :, ::2, ::2] - torch.from_numpy(z["image_half"].astype(np.float32))).abs().max().item() < 5e-3
    # BASELINE config 1: 10-bit tokenizer, encode + decode one 256x256 image on CPU
    z1 = load_golden("tok_full10_cfg1.npz")
    cfg10 = O.TokCfg(token_size=10)
    sd10 = O.make_tokenizer_weights(cfg10, seed=int(z1["seed"]), with_encoder=True)
    assert sha(sd10["encoder.conv_in.weight"]) == str(z1["w_sha_conv_in"])
    x = torch.rand(1, 3, 256, 256, generator=torch.Generator().manual_seed(0))
    _, idx = O.encode_image(sd10, cfg10, x)
    ref_idx = torch.from_numpy(z1["indices"]).long()
    assert (idx != ref_idx).float().mean().item() <= 2 / 256      # sign of a ~0 pre-activation may flip under conv-algo noise
    rec = O.decode_tokens(sd10, cfg10, ref_idx.reshape(1, -1))
    assert (rec[:, :, 100:132, 100:132] - torch.from_numpy(z1["recon_crop"])).abs().max().item() < 2e-4


def test_generator_variants_match_reference_goldens():
    """Oracle restatement of LFQBert(use_prenorm=True) and of the embedding-table Bert (post-/pre-norm, 2 and 3 groups) against
    logits captured from the real reference classes (oracle/make_golden_variants.py)."""
    from oracle.make_golden_variants import VARIANTS
    z = load_golden("gen_variants_tiny.npz")
    for name, (cfg, seed) in VARIANTS.items():
        sd = O.make_generator_weights(cfg, seed=seed, head_gain=20.0)
        assert sha(sd["pos_emb"]) == str(z[f"{name}.w_sha"])
        got = O.lfq_bert_forward(sd, cfg, torch.from_numpy(z[f"{name}.tokens"]), torch.from_numpy(z[f"{name}.labels"]), torch.from_numpy(z[f"{name}.drop"]))
        assert float((got - torch.from_numpy(z[f"{name}.logits"])).abs().max()) < 2e-4, name


def test_return_attn_matches_reference_goldens():
    """return_attn=True (bert.py:461, 505-508): logits unchanged and one head-averaged attention map [b, 257, 257] per layer,
    post- / pre-norm LFQBert and the table Bert, against maps captured from the reference (stored as fp16)."""
    from oracle.make_golden_variants import ATTN_VARIANTS
    z = load_golden("gen_variants_tiny.npz")
    for name, (cfg, seed) in ATTN_VARIANTS.items():
        sd = O.make_generator_weights(cfg, seed=seed, head_gain=20.0)
        assert sha(sd["pos_emb"]) == str(z[f"{name}.w_sha"])
        toks, labels, drop = (torch.from_numpy(z[f"{name}.{k}"]) for k in ("tokens", "labels", "drop"))
        logits, attn = O.lfq_bert_forward(sd, cfg, toks, labels, drop, return_attn=True)
        assert float((logits - torch.from_numpy(z[f"{name}.logits"])).abs().max()) < 2e-4, name
        assert torch.equal(logits, O.lfq_bert_forward(sd, cfg, toks, labels, drop))
        want = torch.from_numpy(z[f"{name}.attn"]).float()
        assert len(attn) == cfg.depth and want.shape == (cfg.depth, 2, 257, 257)
        for l in range(cfg.depth):
            assert float((attn[l] - want[l]).abs().max()) < 1e-3, (name, l)               # fp16 storage of values in [0, 1]
            assert float((attn[l].sum(-1) - 1).abs().max()) < 1e-5


def test_tokenizer_average_pool_variant_matches_reference_golden():
    """sample_with_conv=False: F.avg_pool2d(2, 2) between the encoder stages instead of the stride-2 convolution (autoencoder.py:179-182)."""
    from oracle.make_golden_variants import AVGPOOL_TOK as cfg
    z = load_golden("tok_avgpool_tiny.npz")
    sd = O.make_tokenizer_weights(cfg, seed=int(z["seed"]), with_encoder=True)
    assert sha(sd["encoder.conv_in.weight"]) == str(z["w_sha_enc_conv_in"]) and len(sd) == int(z["n_keys"])
    assert not any("down_conv" in k for k in sd)
    zq, idx = O.encode_image(sd, cfg, torch.from_numpy(z["enc_input"]))
    assert torch.equal(idx, torch.from_numpy(z["enc_indices"])) and torch.equal(zq, torch.from_numpy(z["enc_zq"]).float())
    rec = O.decode_latents(sd, cfg, zq)
    assert float((rec - torch.from_numpy(z["recon"]).float()).abs().max()) < 5e-3          # fp16 storage of O(1) pixels
