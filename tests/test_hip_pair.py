"""Differential classifier-free guidance (mb_gen_cfg.precision >= 1, DESIGN.md "Precision"): the pair GEMM, the pair LayerNorm / attention
operands, and the guided forward against the CPU oracle.  The reference computes the guided logits as
c + s (c - u) from one forward over [cond | uncond] (sampling.py:83-99); the engine carries the unconditional stream's fp16 GEMM operands
as differences from the conditional stream's, so that operand rounding cancels in (c - u)."""
import pytest
import torch

from hip_helpers import hip_generator
from oracle import maskbit_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("epi,pairs,N,K", [(0, 2, 768, 256), (1, 3, 512, 1024), (2, 2, 256, 512), (2, 5, 1024, 1024)])
def test_pair_gemm(epi, pairs, N, K):
    """out_c = f(A_c.W^T + b), out_u = f((A_c + A_delta).W^T + b) with f = identity->fp16 / GELU (u rows: gelu(u) - gelu(c)) / + residual -> fp32,
    against fp64 on the same fp16 operands: rows of whole 257-token sequences, class-token rows included."""
    from maskbit_amd import _lib
    lib = _lib.load()
    torch.manual_seed(epi * 7 + pairs)
    P = pairs * 257
    xc = torch.randn(P, K, device=DEV) * 1.5
    xu = xc + torch.randn(P, K, device=DEV) * 0.05
    A = torch.cat([xc.half(), (xu - xc).half()])
    W = (torch.randn(N, K, device=DEV) * 0.05).half()
    bias = torch.randn(N, device=DEV) * 0.1
    res = torch.randn(2 * P, N, device=DEV) if epi == 2 else None
    out32 = res.clone() if epi == 2 else None                     # in place, as the engine's residual stream
    out16 = torch.full((2 * P, N), float("nan"), device=DEV, dtype=torch.float16) if epi != 2 else None
    st = torch.cuda.current_stream().cuda_stream
    from hip_helpers import gemm_mini
    gemm_mini(lib, epi, A, W, bias, out32, out32, out16, P, True, N, K)
    torch.cuda.synchronize()
    pc = A[:P].double() @ W.double().t() + bias.double()
    pu = (A[:P].double() + A[P:].double()) @ W.double().t() + bias.double()
    if epi == 1:
        gc, gu = torch.nn.functional.gelu(pc), torch.nn.functional.gelu(pu)
        want = torch.cat([gc, gu - gc])
    elif epi == 2:
        want = torch.cat([pc, pu]) + res.double()
    else:
        want = torch.cat([pc, pu])
    got = (out32 if out32 is not None else out16).double()
    assert torch.isfinite(got).all()
    err = (got - want).abs()
    tol = 3e-5 if epi == 2 else 2e-3 * max(1.0, float(want.abs().max()))
    assert float(err.max()) < tol, (float(err.max()), int(err.argmax()) // N, int(err.argmax()) % N)


def _full12():
    cfg = O.GenCfg(bits=12, splits=2)
    sd = O.make_generator_weights(cfg, seed=100, head_gain=12.0)
    return cfg, sd, hip_generator(cfg, sd)


@pytest.mark.timeout(900)
def test_guided_forward_full_size_vs_oracle():
    """forward_cfg at full size (B = 3 pairs, masked / unmasked tokens): (a) with precision = 0 it IS the plain forward over [cond | uncond], bit for
    bit; (b) in differential form (precision 1, 2, 3) both streams stay as close to the fp32 oracle as the plain fp16 forward does, and the
    guided combination c + s (c - u) at s = 6 is several times closer -- the operand rounding no longer reaches (c - u)."""
    cfg, sd, m = _full12()
    g = torch.Generator().manual_seed(3)
    t = torch.randint(0, 65, (3, 256, 2), generator=g)
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
    e_by_mode = {}
    for pair in (1, 2, 3, 4):                                                       # differential form; + weight-correction mini-tiles; + activation-lo mini-tiles (3: out-proj + FFN-up; 4: + FFN-down)
        m.precision = pair
        lg = m.forward_cfg(t.to(DEV), y.to(DEV)).cpu()
        rel = float((lg - ref).norm() / ref.norm())
        e_by_mode[pair] = float((guided(lg) - guided(ref)).abs().mean())
        print(f"precision = {pair}: rel-Frobenius logit error {rel:.2e}; mean |guided logit error| {e_by_mode[pair]:.4f} (plain fp16 forward: {e_plain:.4f})")
        assert rel < 2e-3 and e_by_mode[pair] < 0.6 * e_plain
    assert e_by_mode[2] < 0.8 * e_by_mode[1] and e_by_mode[3] < 0.9 * e_by_mode[2] and e_by_mode[4] < e_by_mode[3]
    # precision 4, batch invariance and determinism of the two-set pair GEMMs / the producers' lo copies: pair 1's logits alone, bit for bit
    m.precision = 4
    full4 = m.forward_cfg(t.to(DEV), y.to(DEV))
    one = m.forward_cfg(t[1:2].to(DEV), y[1:2].to(DEV))
    assert torch.equal(one[0], full4[1]) and torch.equal(one[1], full4[4]) and torch.equal(m.forward_cfg(t.to(DEV), y.to(DEV)), full4)
    # the plain forward() of a precision >= 2 engine carries the weight-correction mini-tiles on every trunk GEMM (and hi + lo LayerNorm outputs): closer to the oracle than single fp16
    m.precision = 2
    w = m(torch.cat([t, t]).to(DEV), torch.cat([y, y]).to(DEV), drop.to(DEV)).cpu()
    e_w, e_0 = float((w - ref).abs().mean()), float((plain.cpu() - ref).abs().mean())
    print(f"plain forward: mean |logit error| single fp16 {e_0:.4f}, with the MX-fp4 weight-rounding correction {e_w:.4f}")
    assert e_w < 0.7 * e_0
    sat = m.saturation_count()
    assert sat == 0, sat
    m.precision = -1


@pytest.mark.parametrize("pairs,heads,d", [(2, 16, 1024), (3, 4, 128)])
def test_pair_attention_vs_fp32_reference(pairs, heads, d):
    """mb_attention_pair (bert.py:84,137 on both CFG streams): conditional rows = softmax(QK^T / sqrt(dh)) V, unconditional rows = the DIFFERENCE to
    their conditional twins, against fp64 attention on the same fp16 q / k / v.  The twin's rows are subtracted in fp32 (registers), not from its fp16
    store, so the difference rows carry the two streams' fp16-probability rounding (up to ~7e-5 absolute here, 3e-4 of the largest output) but no
    fp16 output rounding of either stream."""
    from maskbit_amd import _lib
    lib = _lib.load()
    torch.manual_seed(pairs)
    N, dh = 257, d // heads
    qc = torch.randn(pairs * N, 3 * d, device=DEV) * 0.7
    qu = qc + torch.randn(pairs * N, 3 * d, device=DEV) * 0.02          # the unconditional stream differs a little
    qkv = torch.cat([qc, qu]).half().contiguous()
    out = torch.full((2 * pairs * N, d), float("nan"), device=DEV, dtype=torch.float16)
    _lib.check(lib.mb_attention_pair(qkv.data_ptr(), out.data_ptr(), pairs, N, d, heads, torch.cuda.current_stream().cuda_stream),
               "mb_attention_pair")
    torch.cuda.synchronize()
    x = qkv.double().view(2 * pairs, N, 3, heads, dh).permute(2, 0, 3, 1, 4)          # [3, seq, head, N, dh]
    p = torch.softmax(x[0] @ x[1].transpose(-1, -2) / dh ** 0.5, dim=-1)
    o = (p @ x[2]).permute(0, 2, 1, 3).reshape(2 * pairs * N, d)
    oc, ou = o[: pairs * N], o[pairs * N:]
    assert float((out[: pairs * N].double() - oc).abs().max()) < 2e-3 * float(oc.abs().max())               # fp16 probabilities and stores
    diff = ou - oc
    err = float((out[pairs * N:].double() - diff).abs().max())
    assert err < 6e-4 * float(oc.abs().max()) and err < 2e-2 * float(diff.abs().max()), (err, float(diff.abs().max()), float(oc.abs().max()))
    with pytest.raises(RuntimeError):                                             # head widths other than 32 / 64 are refused (N > 288 runs the streaming pair kernel: test_hip_long_seq.py)
        _lib.check(lib.mb_attention_pair(qkv.data_ptr(), out.data_ptr(), 1, N, d, d // 16, torch.cuda.current_stream().cuda_stream), "mb_attention_pair")


@pytest.mark.parametrize("pairs,N", [(2, 257), (1, 1025)])
def test_pair_attention_e2m1_copy_of_the_conditional_outputs(pairs, N):
    """mb_attention_pair_f4 = the launch the engine issues at precision >= 2 (one-block kernel at N = 257, streaming kernel beyond): the fp16 rows are
    bit for bit those of mb_attention_pair, and out4 / out4_scale hold a valid MX-fp4 quantisation of the CONDITIONAL outputs -- one E8M0 scale per (row,
    head) = 64 values, chosen without saturation, in the lane order the out-projection's mini-tile pass reads; class-token rows take no part."""
    from hip_helpers import f4_block_exponent, f4_decode, f4_scale_index
    from maskbit_amd import _lib
    lib = _lib.load()
    torch.manual_seed(N)
    d, heads = 1024, 16
    qc = torch.randn(pairs * N, 3 * d, device=DEV) * 0.7
    qu = qc + torch.randn(pairs * N, 3 * d, device=DEV) * 0.02
    qkv = torch.cat([qc, qu]).half().contiguous()
    st = torch.cuda.current_stream().cuda_stream
    ref = torch.full((2 * pairs * N, d), float("nan"), device=DEV, dtype=torch.float16)
    _lib.check(lib.mb_attention_pair(qkv.data_ptr(), ref.data_ptr(), pairs, N, d, heads, st), "mb_attention_pair")
    out = torch.full_like(ref, float("nan"))
    G = (N - 1) // 64
    out4 = torch.zeros(2 * pairs * N, 2 * d, device=DEV, dtype=torch.uint8)
    out4s = torch.zeros(heads * pairs * G * 64 + 256, device=DEV, dtype=torch.uint8)
    out4l, out4ls = torch.zeros_like(out4), torch.zeros_like(out4s)
    _lib.check(lib.mb_attention_pair_f4(qkv.data_ptr(), out.data_ptr(), out4.data_ptr(), out4s.data_ptr(), out4l.data_ptr(), out4ls.data_ptr(), pairs, N, d, heads, st),
               "mb_attention_pair_f4")
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    # without the lo copy (precision 2 / 3): the same value copy and rows, bit for bit
    o2, c2, s2 = torch.full_like(ref, float("nan")), torch.zeros_like(out4), torch.zeros_like(out4s)
    _lib.check(lib.mb_attention_pair_f4(qkv.data_ptr(), o2.data_ptr(), c2.data_ptr(), s2.data_ptr(), None, None, pairs, N, d, heads, st), "mb_attention_pair_f4")
    assert torch.equal(o2, out) and torch.equal(c2, out4) and torch.equal(s2, out4s)
    rows = torch.arange(pairs * N)
    seq, tok = rows // N, rows % N
    keep = tok < N - 1
    sb = torch.stack([out4s.cpu()[f4_scale_index(h, pairs, seq[keep], tok[keep], G)] for h in range(heads)], 1).double()      # [rows, heads]
    dec = (f4_decode(out4[: pairs * N][keep.to(DEV)], d).reshape(-1, heads, 64) * (2.0 ** (sb - 127)).unsqueeze(-1)).reshape(-1, d)
    want = out[: pairs * N][keep.to(DEV)].double().cpu()                      # (the copy is taken from the fp32 tile; the fp16 rows are within 2^-11 of it)
    amax = want.reshape(-1, heads, 64).abs().amax(-1)
    E = f4_block_exponent(amax.clamp(min=1e-30))
    assert float((sb != (E - 2).clamp(min=0).double()).double().mean()) < 5e-3          # (fp32 tile vs its fp16 rounding may straddle a binade)
    assert float(((dec - want) ** 2).sum() / (want ** 2).sum()) < 0.02
    assert int(out4[pairs * N:].count_nonzero()) == 0                       # nothing is written for the difference rows
    # the lo copy (precision 4: the out-projection's activation-lo pass): e2m1 of o_c - fp16(o_c) taken from the same fp32 tile.  The tile itself is not
    # visible from here, so: (a) every decoded lo value is a rounding remainder of its fp16 row entry (|lo| <= ulp / 2, up to the e2m1 grid's own rounding);
    # (b) fp16 row + decoded lo is CLOSER to the fp64 attention of the same fp16 q / k / v rows than the fp16 row alone
    sbl = torch.stack([out4ls.cpu()[f4_scale_index(h, pairs, seq[keep], tok[keep], G)] for h in range(heads)], 1).double()
    lo = (f4_decode(out4l[: pairs * N][keep.to(DEV)], d).reshape(-1, heads, 64) * (2.0 ** (sbl - 127)).unsqueeze(-1)).reshape(-1, d)
    ulp = 2.0 ** (torch.floor(torch.log2(want.abs().clamp(min=2.0 ** -14))) - 10)
    assert float((lo.abs() / ulp).max()) <= 0.5 * 1.34 and float((lo != 0).double().mean()) > 0.5
    q, k, v = [t.double().reshape(pairs, N, heads, 64).transpose(1, 2) for t in qkv[: pairs * N].split(d, -1)]
    exact = (torch.softmax(q @ k.transpose(-1, -2) / 8.0, -1) @ v).transpose(1, 2).reshape(pairs * N, d).cpu()[keep]
    e_hi, e_lo = float((want - exact).pow(2).mean().sqrt()), float((want + lo - exact).pow(2).mean().sqrt())
    print(f"attention outputs vs fp64 on the same fp16 q / k / v: fp16 rows {e_hi:.3e}, + decoded e2m1 lo halves {e_lo:.3e}")
    assert e_lo < 0.8 * e_hi
    assert int(out4l[pairs * N:].count_nonzero()) == 0


@pytest.mark.timeout(900)
def test_activation_lo_coverage_knob_and_mode_identities():
    """The precision modes are nested selections of the same machinery, bit for bit (full-width two-layer generator, guided forward of 3 pairs):
    precision 4 with its activation-lo sets switched off (mb_gen_set_alo mask 0) IS precision 2; precision 4 narrowed to out-proj + FFN-up (mask 6) IS precision 3;
    the knob refuses what the handle was not built with; and mb_sample at precision 4 equals the forward + step composition in which EVERY step of a guided run --
    the zero-scale ones too -- takes the guided forward."""
    from maskbit_amd import _lib
    from maskbit_amd.sampling import build_plan, draw_noise, run_loop
    lib = _lib.load()
    cfg = O.GenCfg(bits=12, splits=2, depth=2)
    sd = O.make_generator_weights(cfg, seed=31, head_gain=12.0)
    from hip_helpers import hip_generator
    m = hip_generator(cfg, sd)
    g = torch.Generator().manual_seed(9)
    t = torch.randint(0, 65, (3, 256, 2), generator=g).to(DEV)
    y = torch.tensor([5, 321, 999], device=DEV)
    out = {}
    for prec in (2, 3, 4):
        m.precision, m.alo_mask = prec, None
        out[prec] = m.forward_cfg(t, y)
    assert not torch.equal(out[2], out[3]) and not torch.equal(out[3], out[4])
    m.precision, m.alo_mask, m.alo_from = 4, 0, 0
    assert torch.equal(m.forward_cfg(t, y), out[2])
    m.alo_mask = 6
    assert torch.equal(m.forward_cfg(t, y), out[3])
    m.alo_mask = 14
    assert torch.equal(m.forward_cfg(t, y), out[4])
    m.alo_mask = 15                                                  # the QKV set: built at precision 4 for coverage studies, not run by default
    assert not torch.equal(m.forward_cfg(t, y), out[4])
    m.precision, m.alo_mask = 3, 14                                  # a precision-3 handle holds the operands of out-proj + FFN-up only
    with pytest.raises(RuntimeError, match="created"):
        m.forward_cfg(t, y)
    # mb_sample at precision 4 == composition with the guided forward at every step (cosine annealing: the scale of steps 0 .. 2 is exactly 0)
    m.precision, m.alo_mask = 4, None
    N, B = 8, 3
    plan = build_plan(N, 512, 7.1, "cosine", 3.0, 1.0, False, "arccos")
    assert plan[0][0] == 0.0 and plan[0][-1] != 0.0
    torch.manual_seed(3)
    q, c = draw_noise(B, 256, 2, 64, N, 8.2, torch.device(DEV))
    _, _, steps, _ = run_loop(m, None, y, plan, q, c, want_image=False)
    tok = torch.full((B, 256, 2), 64, dtype=torch.int64, device=DEV)
    for i in range(N):
        lg = m.forward_cfg(tok, y)
        lc, lu = lg[:B].contiguous(), lg[B:].contiguous()
        tout, pred = torch.empty_like(tok), torch.empty_like(tok)
        _lib.check(lib.mb_sample_step(lc.data_ptr(), lu.data_ptr(), plan[0][i], plan[1][i], q[i].data_ptr(), c[i].data_ptr(), plan[2][i], tok.data_ptr(),
                                      tout.data_ptr(), pred.data_ptr(), B, 256, 2, 64, torch.cuda.current_stream().cuda_stream))
        assert torch.equal(pred, steps[i]), f"step {i}"
        tok = tout
    m.precision = -1


@pytest.mark.timeout(900)
def test_guided_and_plain_forward_at_hidden_768():
    """The other trunk width the pair / mini tiles serve (hidden 768 = the reference constructor's default width, 12 heads of 64, mlp 3072; K = 768 = six
    mini-tiles per operand set, N = 768 / 2 304 / 3 072: three / nine / twelve 256-column tiles): guided forward at every precision against the fp32 oracle --
    each step closer in the guided combination --, pair batch invariance, the plain forward with and without the weight correction."""
    from hip_helpers import hip_generator
    cfg = O.GenCfg(bits=12, splits=2, hidden=768, depth=2, heads=12, mlp=3072)
    sd = O.make_generator_weights(cfg, seed=41, head_gain=12.0)
    m = hip_generator(cfg, sd)
    assert m.resolved_precision() == 2
    g = torch.Generator().manual_seed(11)
    t = torch.randint(0, 65, (3, 256, 2), generator=g)
    y = torch.tensor([5, 321, 999])
    drop = torch.cat([torch.zeros(3, dtype=torch.bool), torch.ones(3, dtype=torch.bool)])
    ref = O.lfq_bert_forward(sd, cfg, torch.cat([t, t]), torch.cat([y, y]), drop)
    guided = lambda lg: lg[:3] + 6.0 * (lg[:3] - lg[3:])
    m.precision = 0
    plain0 = m(torch.cat([t, t]).to(DEV), torch.cat([y, y]).to(DEV), drop.to(DEV))
    assert torch.equal(m.forward_cfg(t.to(DEV), y.to(DEV)), plain0)
    e = {0: float((guided(plain0.cpu()) - guided(ref)).abs().mean())}
    for prec in (1, 2, 3, 4):
        m.precision = prec
        lg = m.forward_cfg(t.to(DEV), y.to(DEV))
        assert float((lg.cpu() - ref).norm() / ref.norm()) < 2e-3
        e[prec] = float((guided(lg.cpu()) - guided(ref)).abs().mean())
        one = m.forward_cfg(t[1:2].to(DEV), y[1:2].to(DEV))
        assert torch.equal(one[0], lg[1]) and torch.equal(one[1], lg[4])
    print(f"hidden 768: mean |guided logit error| by precision {e}")
    assert e[1] < 0.6 * e[0] and e[2] < 0.85 * e[1] and e[3] < e[2] and e[4] < e[3]
    m.precision = 2
    w = m(torch.cat([t, t]).to(DEV), torch.cat([y, y]).to(DEV), drop.to(DEV)).cpu()
    e_w, e_0 = float((w - ref).abs().mean()), float((plain0.cpu() - ref).abs().mean())
    print(f"hidden 768, plain forward: mean |logit error| single fp16 {e_0:.4f}, default {e_w:.4f}")
    assert e_w < 0.75 * e_0 and m.saturation_count() == 0
    m.precision = -1
