"""512x512 models: 32x32 = 1024 image tokens + the class token.  One head's K/V no longer fits in LDS, so
attention runs the streaming kernel (128-key blocks, online softmax); everything else is the same code at a larger N."""
import pytest
import torch

from hip_helpers import f4_encode_rows, gemm_mini, hip_tokenizer, w4_decode
from oracle import maskbit_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _gen(cfg, sd):
    from maskbit_amd import LFQBert
    m = LFQBert(img_size=512, hidden_dim=cfg.hidden, codebook_size=2 ** cfg.bits, codebook_splits=cfg.splits, depth=cfg.depth, heads=cfg.heads,
                mlp_dim=cfg.mlp, dropout=0.1, nclass=cfg.nclass, input_stride=16)
    assert m.seq_len == 1024
    m.load_state_dict(sd, strict=True)
    return m.eval().requires_grad_(False).to(DEV)


@pytest.mark.parametrize("hidden,heads,depth,b", [(128, 4, 2, 3), (128, 2, 1, 2), (1024, 16, 1, 2)])
def test_generator_1025_tokens_vs_oracle(hidden, heads, depth, b):
    """head dims 32 and 64; 1025 = 8 key blocks + 1 key: the last block is almost entirely masked"""
    cfg = O.GenCfg(bits=12, splits=2, hidden=hidden, depth=depth, heads=heads, mlp=2 * hidden, seq=1024, nclass=10)
    sd = O.make_generator_weights(cfg, seed=hidden + heads, head_gain=12.0)
    m = _gen(cfg, sd)
    g = torch.Generator().manual_seed(b)
    toks = torch.randint(0, 65, (b, 1024, 2), generator=g); y = torch.randint(0, 10, (b,), generator=g)
    drop = torch.tensor([False, True, False][:b])
    out = m(toks.to(DEV), y.to(DEV), drop.to(DEV))
    ref = O.lfq_bert_forward(sd, cfg, toks, y, drop)
    rel = float((out.cpu() - ref).norm() / ref.norm())
    print(f"hidden {hidden} heads {heads}: rel-Frobenius logit error {rel:.2e}")
    assert out.shape == (b, 1024, 2, 64) and torch.isfinite(out).all() and rel < 2e-3
    assert torch.equal(m(toks[:1].to(DEV), y[:1].to(DEV), drop[:1].to(DEV)), out[:1])      # batch invariance


def test_sample_512_end_to_end_tiny():
    """sample() with a 1024-token generator and a 32x32-latent decode to 128x128 (tiny tokenizer, 3 resolutions)."""
    from maskbit_amd import sample
    gcfg = O.GenCfg(bits=12, splits=2, hidden=128, depth=1, heads=4, mlp=256, seq=1024, nclass=10)
    gsd = O.make_generator_weights(gcfg, seed=5, head_gain=12.0)
    m = _gen(gcfg, gsd)
    tcfg = O.TokCfg(token_size=12, hidden_channels=64, channel_mult=(1, 1, 2), num_resolutions=3, num_res_blocks=1)
    tsd = O.make_tokenizer_weights(tcfg, seed=21, with_encoder=True)
    tk = hip_tokenizer(tcfg, tsd)
    torch.manual_seed(0)
    img, steps = sample(m, tk, num_samples=2, labels=torch.tensor([1, 7]).to(DEV), num_steps=4, guidance_scale=2.0, mask_token=64, patch_size=32,
                        codebook_size=4096, codebook_splits=2, randomize_temperature=2.0, mask_schedule_strategy="arccos")
    assert img.shape == (2, 3, 128, 128) and torch.isfinite(img).all()
    assert len(steps) == 4 and steps[-1].shape == (2, 1024, 2) and int(steps[-1].max()) < 64
    codes = O.combine_groups(steps[-1].cpu(), 12, 2)
    want = O.decode_tokens(tsd, tcfg, codes)
    assert float((img.cpu() - want).abs().max()) < 0.03


# ---- round 5: the DIFFERENTIAL guided forward of the 1024 + 1-token models (eval_maskbit.py:125,139-144 -> bert.py:377,390) ------------------------
@pytest.mark.parametrize("epi,pairs,N,K,nlo", [(0, 1, 768, 1024, 1), (1, 2, 512, 1024, 1), (2, 1, 256, 2048, 1), (0, 1, 512, 1024, 2), (2, 2, 1024, 1024, 0)])
def test_pair_gemm_1025_token_sequences(epi, pairs, N, K, nlo):
    """Pair tiles over sequences of 1 025 rows (mb_gemm_mini_seq): eight 128-token tiles per sequence pair, the class rows stored by the last one, the
    token operand's scale bytes in 16 groups of 64 tokens per sequence -- against fp64 on the same (decoded) operands, class-token rows included."""
    from maskbit_amd import _lib
    lib = _lib.load()
    torch.manual_seed(epi + pairs + nlo + K)
    SQ = 1025
    P = pairs * SQ
    xc = torch.randn(P, K, device=DEV) * (0.2 + torch.rand(P, K // 64, device=DEV).repeat_interleave(64, 1) * 3)
    xu = xc + torch.randn(P, K, device=DEV) * 0.05
    A = torch.cat([xc.half(), (xu - xc).half()])
    W32 = torch.randn(N, K, device=DEV) * 0.03 * (0.5 + torch.rand(N, 1, device=DEV) * 2)
    W = W32.half()
    st = torch.cuda.current_stream().cuda_stream
    sets, corr = [], 0
    if nlo >= 1:
        w4 = torch.zeros(N, 2 * K, device=DEV, dtype=torch.uint8); ws = torch.zeros(N * K // 128, device=DEV, dtype=torch.uint8)
        _lib.check(lib.mb_w4lo_from_f32(W32.data_ptr(), N, K, w4.data_ptr(), ws.data_ptr(), st))
        x4, xs, x4_dec = f4_encode_rows(A[:P].double(), pairs, seq_rows=SQ)
        sets.append((x4, xs, w4, ws))
        corr = x4_dec @ w4_decode(w4, ws, N, K).t()
    if nlo == 2:
        w4v = torch.zeros(N, 2 * K, device=DEV, dtype=torch.uint8); wsv = torch.zeros(N * K // 128, device=DEV, dtype=torch.uint8)
        _lib.check(lib.mb_w4_from_f32(W32.data_ptr(), N, K, w4v.data_ptr(), wsv.data_ptr(), st))
        xl4, xls, xl_dec = f4_encode_rows(xc.double() - A[:P].double(), pairs, seq_rows=SQ)
        sets.append((xl4, xls, w4v, wsv))
        corr = corr + xl_dec @ w4_decode(w4v, wsv, N, K).t()
    bias = torch.randn(N, device=DEV) * 0.1
    res = torch.randn(2 * P, N, device=DEV) if epi == 2 else None
    out32 = res.clone() if epi == 2 else None
    out16 = torch.full((2 * P, N), float("nan"), device=DEV, dtype=torch.float16) if epi != 2 else None
    gemm_mini(lib, epi, A, W, bias, out32, out32, out16, P, True, N, K, sets, seq_rows=SQ)
    torch.cuda.synchronize()
    pc = A[:P].double() @ W.double().t() + corr + bias.double()
    pu = pc + A[P:].double() @ W.double().t()
    if epi == 1:
        gc, gu = torch.nn.functional.gelu(pc), torch.nn.functional.gelu(pu)
        want = torch.cat([gc, gu - gc])
    else:
        want = torch.cat([pc, pu]) + (res.double() if res is not None else 0)
    got = (out32 if out32 is not None else out16).double()
    assert torch.isfinite(got).all()
    err = (got - want).abs()
    tol = 6e-5 if epi == 2 else 2e-3 * max(1.0, float(want.abs().max()))
    assert float(err.max()) < tol, (float(err.max()), int(err.argmax()) // N, int(err.argmax()) % N)


@pytest.mark.parametrize("pairs,heads,d,N", [(1, 16, 1024, 1025), (2, 4, 128, 1025), (2, 2, 128, 300)])
def test_pair_attention_long_sequences(pairs, heads, d, N):
    """mb_attention_pair beyond one head's K / V in LDS (N > 288: the streaming kernel, both streams of a pair in one workgroup): conditional rows =
    softmax(QK^T / sqrt(dh)) V, unconditional rows = the difference to their conditional twins taken in fp32, against fp64 on the same fp16 q / k / v."""
    from maskbit_amd import _lib
    lib = _lib.load()
    torch.manual_seed(pairs + N)
    dh = d // heads
    qc = torch.randn(pairs * N, 3 * d, device=DEV) * 0.7
    qu = qc + torch.randn(pairs * N, 3 * d, device=DEV) * 0.02
    qkv = torch.cat([qc, qu]).half().contiguous()
    out = torch.full((2 * pairs * N, d), float("nan"), device=DEV, dtype=torch.float16)
    _lib.check(lib.mb_attention_pair(qkv.data_ptr(), out.data_ptr(), pairs, N, d, heads, torch.cuda.current_stream().cuda_stream), "mb_attention_pair")
    torch.cuda.synchronize()
    x = qkv.double().view(2 * pairs, N, 3, heads, dh).permute(2, 0, 3, 1, 4)
    p = torch.softmax(x[0] @ x[1].transpose(-1, -2) / dh ** 0.5, dim=-1)
    o = (p @ x[2]).permute(0, 2, 1, 3).reshape(2 * pairs * N, d)
    oc, ou = o[: pairs * N], o[pairs * N:]
    assert torch.isfinite(out.float()).all()
    assert float((out[: pairs * N].double() - oc).abs().max()) < 2e-3 * float(oc.abs().max())
    diff = ou - oc
    err = float((out[pairs * N:].double() - diff).abs().max())
    assert err < 6e-4 * float(oc.abs().max()) and err < 3e-2 * float(diff.abs().max()), (err, float(diff.abs().max()), float(oc.abs().max()))


@pytest.mark.timeout(900)
def test_guided_forward_1025_tokens_full_width_vs_oracle():
    """forward_cfg of a full-width (hidden 1024, 16 heads, mlp 4096) two-layer generator over 1024 + 1 tokens: the product default now resolves to the
    differential form with the weight-correction mini-tiles (precision 2) here too; with precision = 0 it is the plain forward over [cond | uncond] bit
    for bit; in differential form the guided combination at s = 6 is several times closer to the fp32 oracle; batch invariance of a pair."""
    cfg = O.GenCfg(bits=12, splits=2, hidden=1024, depth=2, heads=16, mlp=4096, seq=1024, nclass=1000)
    sd = O.make_generator_weights(cfg, seed=77, head_gain=12.0)
    m = _gen(cfg, sd)
    assert m.resolved_precision() == 2
    g = torch.Generator().manual_seed(5)
    t = torch.randint(0, 65, (3, 1024, 2), generator=g)
    y = torch.tensor([5, 321, 999])
    drop = torch.cat([torch.zeros(3, dtype=torch.bool), torch.ones(3, dtype=torch.bool)])
    torch.set_num_threads(min(16, torch.get_num_threads()))
    ref = O.lfq_bert_forward(sd, cfg, torch.cat([t, t]), torch.cat([y, y]), drop)
    s = 6.0
    guided = lambda lg: lg[:3] + s * (lg[:3] - lg[3:])
    m.precision = 0
    plain = m(torch.cat([t, t]).to(DEV), torch.cat([y, y]).to(DEV), drop.to(DEV))
    assert torch.equal(m.forward_cfg(t.to(DEV), y.to(DEV)), plain)
    e_plain = float((guided(plain.cpu()) - guided(ref)).abs().mean())
    es = {}
    for pair in (1, 2, 4):                                  # (4, round 6: activation-lo sets on every GEMM; the streaming pair attention writes the lo copy too)
        m.precision = pair
        lg = m.forward_cfg(t.to(DEV), y.to(DEV))
        rel = float((lg.cpu() - ref).norm() / ref.norm())
        e = es[pair] = float((guided(lg.cpu()) - guided(ref)).abs().mean())
        print(f"1025 tokens, precision = {pair}: rel-Frobenius logit error {rel:.2e}; mean |guided logit error| {e:.4f} (plain fp16 forward over [cond | uncond]: {e_plain:.4f})")
        assert rel < 2e-3 and e < 0.6 * e_plain
        one = m.forward_cfg(t[1:2].to(DEV), y[1:2].to(DEV))
        assert torch.equal(one[0], lg[1]) and torch.equal(one[1], lg[4])
    assert es[4] < 0.9 * es[2] < 0.9 * es[1]
    assert m.saturation_count() == 0
    m.precision = -1


@pytest.mark.timeout(900)
def test_plain_forward_1025_tokens_precision_modes_full_width_vs_oracle():
    """forward() of a full-width two-layer generator over 1024 + 1 tokens under the precision knob (round 5: the plain sequence tiles now serve 1 025-row
    sequences as four 256-token tiles, so precision 2 means the same here as for the 257-token models): 1 = hi + lo LayerNorm outputs in FFN-up, 2 = + the
    MX-fp4 weight-correction mini-tiles on every trunk GEMM (the streaming attention kernel writes the e2m1 copy of its outputs for the out-projection).
    Each step moves the logits closer to the fp32 oracle; batch invariance bit for bit -- 1 sequence (quarter-column tiles), 3, and 17 (whole tiles)."""
    cfg = O.GenCfg(bits=12, splits=2, hidden=1024, depth=2, heads=16, mlp=4096, seq=1024, nclass=1000)
    sd = O.make_generator_weights(cfg, seed=78, head_gain=12.0)
    m = _gen(cfg, sd)
    g = torch.Generator().manual_seed(6)
    t = torch.randint(0, 65, (3, 1024, 2), generator=g)
    y = torch.tensor([7, 123, 998])
    drop = torch.tensor([False, True, False])
    torch.set_num_threads(min(16, torch.get_num_threads()))
    ref = O.lfq_bert_forward(sd, cfg, t, y, drop)
    err = {}
    for prec in (0, 1, 2):
        m.precision = prec
        out = m(t.to(DEV), y.to(DEV), drop.to(DEV))
        err[prec] = float((out.cpu() - ref).norm() / ref.norm())
        assert torch.isfinite(out).all()
        assert torch.equal(m(t[1:2].to(DEV), y[1:2].to(DEV), drop[1:2].to(DEV)), out[1:2])
        big = m(t.repeat(6, 1, 1)[:17].to(DEV), y.repeat(6)[:17].to(DEV), drop.repeat(6)[:17].to(DEV))
        assert torch.equal(big[:3], out) and torch.equal(big[15:17], out[:2])
    print(f"1025 tokens, plain forward: rel-Frobenius logit error single fp16 {err[0]:.2e}, hi + lo LayerNorm outputs {err[1]:.2e}, + weight correction {err[2]:.2e}")
    assert err[1] < 0.995 * err[0] and err[2] < 0.75 * err[1]
    assert m.saturation_count() == 0
