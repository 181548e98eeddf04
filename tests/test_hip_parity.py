"""GPU parity tests: the HIP engine (through the C ABI) against the CPU oracle and the reference's
golden vectors.  Floating-point tolerances are stated per test; the storage type is fp16 with fp32
accumulation, the reference is fp32 end to end.  The sampling STEP is expected to match bit for bit
(all of its arithmetic is fp32)."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

from conftest import load_golden, golden_weights
from hip_helpers import hip_generator, hip_tokenizer
from oracle import maskbit_oracle as O

pytestmark = pytest.mark.gpu

TINY_GEN = O.GenCfg(bits=12, splits=2, hidden=128, depth=2, heads=4, mlp=256, seq=256, nclass=10)
TINY_TOK = O.TokCfg(token_size=12, hidden_channels=64, channel_mult=(1, 1, 2), num_resolutions=3, num_res_blocks=1)
DEV = "cuda"


def rel_fro(got, ref):
    return float((got.float().cpu() - ref).norm() / ref.norm())


# --------------------------------------------------------------------------------------------- generator
def test_generator_forward_tiny_vs_reference_golden():
    z = load_golden("gen_tiny.npz")
    m = hip_generator(TINY_GEN, golden_weights(z))
    lab = torch.from_numpy(z["labels"]).to(DEV)
    lab0 = lab.clone()
    out = m(torch.from_numpy(z["tokens"]).to(DEV), lab, torch.from_numpy(z["drop"]).to(DEV))
    ref = torch.from_numpy(z["logits"])
    assert out.shape == ref.shape and out.dtype == torch.float32 and out.device.type == "cuda"
    assert torch.equal(lab, lab0)                                   # labels are not mutated
    assert rel_fro(out, ref) < 2e-3                                 # fp16 storage: measured ~4e-4
    assert float((out.cpu() - ref).abs().max()) < 0.004 * float(ref.abs().max())


def test_generator_forward_full12_vs_reference_golden():
    z = load_golden("gen_full12.npz")
    cfg = O.GenCfg(bits=12, splits=2)
    sd = O.make_generator_weights(cfg, seed=int(z["seed"]), head_gain=float(z["head_gain"]))
    m = hip_generator(cfg, sd)
    out = m(torch.from_numpy(z["tokens"]).to(DEV), torch.from_numpy(z["labels"]).to(DEV), torch.from_numpy(z["drop"]).to(DEV))
    ref = torch.from_numpy(z["logits"])
    assert rel_fro(out, ref) < 4e-3                                 # 24 layers of fp16-rounded GEMM operands: measured ~1.2e-3
    agree = (torch.softmax(ref, -1).argmax(-1) == torch.softmax(out.cpu(), -1).argmax(-1)).float().mean()
    assert float(agree) > 0.995


def test_generator_plain_forward_precision_modes_full12_and_tiny():
    """The plain forward() under the engine's ONE precision knob (LFQBert.precision): 0 = single fp16 operands; 1 = + the LayerNorm output in front of
    FFN-up as an fp16 hi + lo pair (the GEMM-input rounding is 13 % of the error variance); 2 = + the MX-fp4 weight-correction mini-tiles on every
    trunk GEMM (the weights' rounding is most of the rest).  Each step moves the logits closer to the reference's fp32 golden; batch invariance holds
    bit for bit in every mode; switching the mode on a live model rebuilds the engine.  The tiny model (128-square GEMM kernel, no pair / mini tiles)
    runs modes >= 1 with hi + lo LayerNorm outputs alone -- in QKV and FFN-up of every layer (no weight correction there to carry the margin)."""
    z = load_golden("gen_full12.npz")
    cfg = O.GenCfg(bits=12, splits=2)
    sd = O.make_generator_weights(cfg, seed=int(z["seed"]), head_gain=float(z["head_gain"]))
    m = hip_generator(cfg, sd)
    args = (torch.from_numpy(z["tokens"]).to(DEV), torch.from_numpy(z["labels"]).to(DEV), torch.from_numpy(z["drop"]).to(DEV))
    ref = torch.from_numpy(z["logits"])
    err = {}
    for prec in (0, 1, 2):
        m.precision = prec                                                                     # independent of MASKBIT_AMD_PRECISION in the environment
        out = m(*args)
        err[prec] = rel_fro(out, ref)
        assert torch.equal(m(args[0][1:2], args[1][1:2], args[2][1:2]), out[1:2])               # batch invariance in this mode too
        rep = m(args[0].repeat(9, 1, 1), args[1].repeat(9), args[2].repeat(9))                  # half-tile kernel (M >= 512)
        nb = args[0].shape[0]
        assert torch.equal(rep[:nb], rep[-nb:]) and abs(rel_fro(rep[:nb], ref) - err[prec]) < 2e-5
    print(f"rel-Frobenius logit error: single fp16 {err[0]:.2e}, hi + lo LayerNorm outputs {err[1]:.2e}, + weight correction {err[2]:.2e}")
    assert err[1] < 0.995 * err[0] and err[2] < 0.75 * err[1]
    m.precision = 0
    assert abs(rel_fro(m(*args), ref) - err[0]) < 1e-9
    m.precision = -1
    assert m.resolved_precision() == 2
    zt = load_golden("gen_tiny.npz")
    mt = hip_generator(TINY_GEN, golden_weights(zt))
    t, y, d = torch.from_numpy(zt["tokens"]).to(DEV), torch.from_numpy(zt["labels"]).to(DEV), torch.from_numpy(zt["drop"]).to(DEV)
    rt = torch.from_numpy(zt["logits"])
    assert mt.resolved_precision() == 1                           # hidden 128: neither pair tiles nor mini-tiles
    mt.precision = 0
    a0 = rel_fro(mt(t, y, d), rt)
    mt.precision = 1
    a1 = rel_fro(mt(t, y, d), rt)
    print(f"tiny: {a0:.2e} -> {a1:.2e}")
    assert a1 < 2e-3 and a1 < 0.97 * a0                           # hi + lo LayerNorm outputs in QKV and FFN-up of every layer: a strict improvement


def test_generator_batch_invariance_and_determinism():
    """Size-independent properties: a sequence's logits do not depend on its batch neighbours, and two runs are bit-identical."""
    z = load_golden("gen_tiny.npz")
    m = hip_generator(TINY_GEN, golden_weights(z))
    t, y, d = torch.from_numpy(z["tokens"]).to(DEV), torch.from_numpy(z["labels"]).to(DEV), torch.from_numpy(z["drop"]).to(DEV)
    full = m(t, y, d)
    again = m(t, y, d)
    assert torch.equal(full, again)
    one = m(t[2:3], y[2:3], d[2:3])
    assert torch.equal(one[0], full[2])
    big = m(t.repeat(7, 1, 1), y.repeat(7), d.repeat(7))            # 35 sequences: crosses GEMM tile boundaries
    assert torch.equal(big[:5], full) and torch.equal(big[30:], full)


def test_generator_batch_invariance_full_size():
    """Size-independent property at BASELINE's full size: a sequence's logits are bit-identical whatever the batch it rides in
    (1 .. 168 sequences: the 128x128 kernel, general and sequence-aligned half-tile GEMMs incl. their quarter- and half-column tiles for small
    batches, ragged CU rounds, engine regrowth)."""
    cfg = O.GenCfg(bits=12, splits=2)
    sd = O.make_generator_weights(cfg, seed=100, head_gain=12.0)
    m = hip_generator(cfg, sd)
    g = torch.Generator().manual_seed(1)
    tok = torch.randint(0, 65, (128, 256, 2), generator=g).to(DEV)
    y = torch.randint(0, 1000, (128,), generator=g).to(DEV)
    drop = (torch.rand(128, generator=g) < 0.5).to(DEV)
    full = m(tok, y, drop)
    assert torch.isfinite(full).all()
    for b in (1, 2, 3, 7, 20, 33, 64, 127):            # (<= 16 / <= 32 sequences: quarter- / half-column tiles in the N = 1024 GEMMs)
        assert torch.equal(m(tok[:b], y[:b], drop[:b]), full[:b]), b
    big = m(torch.cat([tok, tok[:40]]), torch.cat([y, y[:40]]), torch.cat([drop, drop[:40]]))
    assert torch.equal(big[:128], full) and torch.equal(big[128:], full[:40])


def test_generator_rejects_bad_input():
    z = load_golden("gen_tiny.npz")
    m = hip_generator(TINY_GEN, golden_weights(z))
    with pytest.raises(ValueError):
        m(torch.zeros(1, 255, 2, dtype=torch.long, device=DEV), torch.zeros(1, dtype=torch.long, device=DEV))
    with pytest.raises(ValueError):
        m(torch.zeros(1, 256, 2, dtype=torch.long, device=DEV), torch.zeros(2, dtype=torch.long, device=DEV), None, return_attn=True)


# --------------------------------------------------------------------------------------------- sampling step
def _oracle_steps(num_steps, **kw):
    gsd = golden_weights(load_golden("gen_tiny.npz"))
    rec = []
    torch.manual_seed(77)
    fwd = lambda tk, yy, dd: O.lfq_bert_forward(gsd, TINY_GEN, tk, yy, dd)
    O.sample_loop(fwd, 3, torch.tensor([1, 4, 8]), num_steps=num_steps, mask_token=64, codebook_splits=2, record=rec, **kw)
    return rec


@pytest.mark.parametrize("kw", [
    dict(guidance_scale=7.1, guidance_annealing="cosine", scale_pow=3.0, randomize_temperature=8.2, mask_schedule_strategy="arccos"),
    dict(guidance_scale=0.0, randomize_temperature=4.5, mask_schedule_strategy="linear"),
    dict(guidance_scale=3.0, guidance_annealing="linear", randomize_temperature=2.0, mask_schedule_strategy="cosine", use_sampling_annealing=True),
    # the demo's call site (demo_utils.py:139-157): guidance on with annealing "none" -- the full scale at every step, step 0 included
    dict(guidance_scale=3.0, guidance_annealing="none", scale_pow=1.0, randomize_temperature=4.5, mask_schedule_strategy="arccos"),
])
def test_sample_step_bit_exact_vs_oracle(kw):
    from maskbit_amd import _lib
    lib = _lib.load()
    N = 6
    rec = _oracle_steps(N, **kw)
    for i, r in enumerate(rec):
        B, n, m_ = r.tokens_in.shape
        tin = r.tokens_in.to(DEV).contiguous()
        tout, pred = torch.empty_like(tin), torch.empty_like(tin)
        lc = r.logits_c.to(DEV).contiguous()
        lu = r.logits_u.to(DEV).contiguous() if r.logits_u is not None else None
        qn, cn = r.exp_noise.to(DEV).contiguous(), r.conf_noise.to(DEV).contiguous()
        temp = 0.5 + 0.8 * (1 - (i + 1) / N) if kw.get("use_sampling_annealing") else 1.0
        k = int(torch.floor(torch.tensor(r.mask_ratio) * (n * m_)))
        _lib.check(lib.mb_sample_step(lc.data_ptr(), lu.data_ptr() if lu is not None else None, r.scale, temp, qn.data_ptr(), cn.data_ptr(), k,
                                      tin.data_ptr(), tout.data_ptr(), pred.data_ptr(), B, n, m_, 64, torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        assert torch.equal(pred.cpu(), r.pred), f"step {i}: pred differs at {int((pred.cpu() != r.pred).sum())} positions"
        assert torch.equal(tout.cpu(), r.tokens_out), f"step {i}: re-mask differs"


def test_sample_step_rejects_aliasing_and_edge_sizes():
    from maskbit_amd import _lib
    lib = _lib.load()
    t = torch.full((1, 512), 64, dtype=torch.int64, device=DEV)
    lg = torch.zeros(1, 512, 64, device=DEV)
    q = torch.ones(512, 64, device=DEV)
    cn = torch.zeros(1, 512, device=DEV)
    rc = lib.mb_sample_step(lg.data_ptr(), None, 0.0, 1.0, q.data_ptr(), cn.data_ptr(), 5, t.data_ptr(), t.data_ptr(), None, 1, 256, 2, 64,
                            torch.cuda.current_stream().cuda_stream)
    assert rc != 0 and b"alias" in lib.mb_last_error()
    # all-equal confidences (ties): exactly the reference's `conf <= sorted[k-1]` semantics -> everything re-masked
    out = torch.empty_like(t)
    _lib.check(lib.mb_sample_step(lg.data_ptr(), None, 0.0, 1.0, q.data_ptr(), cn.data_ptr(), 5, t.data_ptr(), out.data_ptr(), None, 1, 256, 2, 64,
                                  torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert bool((out == 64).all())


# --------------------------------------------------------------------------------------------- decoder
def test_decoder_tiny_vs_reference_golden():
    z = load_golden("tok_tiny.npz")
    sd = O.make_tokenizer_weights(TINY_TOK, seed=int(z["seed"]), with_encoder=True)
    tk = hip_tokenizer(TINY_TOK, sd)
    img = tk.decode_tokens(torch.from_numpy(z["tokens"]).to(DEV).float())
    ref = torch.from_numpy(z["image"])
    assert img.shape == ref.shape and img.dtype == torch.float32
    assert rel_fro(img, ref) < 5e-3 and float((img.cpu() - ref).abs().max()) < 0.02       # fp16 activations, measured ~1e-3 / 5e-3
    # decode(z) with +-1 latents goes through the same path
    zq = O.index_to_bits(torch.from_numpy(z["tokens"]), 12).reshape(3, 16, 16, 12).permute(0, 3, 1, 2).contiguous()
    assert torch.equal(tk.decode(zq.to(DEV)), img)


def test_decoder_full12_vs_reference_golden_and_uint8():
    z = load_golden("tok_full12.npz")
    cfg = O.TokCfg(token_size=12)
    sd = O.make_tokenizer_weights(cfg, seed=int(z["seed"]), with_encoder=True)
    tk = hip_tokenizer(cfg, sd)
    img, u8 = tk.decode_tokens_uint8(torch.from_numpy(z["tokens"]).to(DEV))
    for (y, x) in ((0, 0), (120, 120), (240, 240), (37, 201)):
        ref = torch.from_numpy(z[f"crop_{y}_{x}"])
        assert float((img[:, :, y:y + 16, x:x + 16].cpu() - ref).abs().max()) < 0.03     # |pixel| up to ~3; measured ~8e-3
    half = torch.from_numpy(z["image_half"].astype(np.float32))
    assert rel_fro(img[:, :, ::2, ::2], half) < 6e-3
    assert np.allclose(img.mean((0, 2, 3)).cpu().numpy(), z["mean"], atol=2e-3)
    want = (torch.clamp(img, 0.0, 1.0) * 255.0).permute(0, 2, 3, 1).to(torch.uint8)       # eval_maskbit.py:134-135 (truncating cast)
    assert torch.equal(u8, want)
    again, _ = tk.decode_tokens_uint8(torch.from_numpy(z["tokens"]).to(DEV))
    assert torch.equal(again, img)                                                         # deterministic GroupNorm reduction


def test_decoder_config1_10bit():
    """BASELINE config 1 (10-bit tokenizer): decode the reference's own encoder indices on the GPU."""
    z = load_golden("tok_full10_cfg1.npz")
    cfg = O.TokCfg(token_size=10)
    sd = O.make_tokenizer_weights(cfg, seed=int(z["seed"]), with_encoder=True)
    tk = hip_tokenizer(cfg, sd)
    rec = tk.decode_tokens(torch.from_numpy(z["indices"]).reshape(1, -1).to(DEV))
    assert float((rec[:, :, 100:132, 100:132].cpu() - torch.from_numpy(z["recon_crop"])).abs().max()) < 0.03
    # (the encode half of this configuration: tests/test_hip_encoder.py)


def test_decoder_fp16_saturation_is_counted():
    """The decoder keeps activations in fp16 with saturating stores: values that leave +-65504 are clamped AND counted, so a
    checkpoint that needs more range is noticed.  Seeded weights: 0 clamps; the same weights with conv_in scaled by 1e6: clamps reported."""
    tz = load_golden("tok_tiny.npz")
    tsd = O.make_tokenizer_weights(TINY_TOK, seed=int(tz["seed"]), with_encoder=True)
    tm = hip_tokenizer(TINY_TOK, tsd)
    toks = torch.from_numpy(tz["tokens"]).to(DEV)
    tm.decode_tokens(toks)
    assert tm.saturation_count() == 0
    big = dict(tsd)
    big["decoder.conv_in.weight"] = tsd["decoder.conv_in.weight"] * 1e6
    tb = hip_tokenizer(TINY_TOK, big)
    out = tb.decode_tokens(toks)
    n = tb.saturation_count()
    assert n > 0 and torch.isfinite(out).all()          # clamped, never inf / NaN
    assert tb.saturation_count() == 0                   # reading resets the counter
