"""The MX-fp4 mini-tile passes of the sequence-aligned trunk GEMM (gemm_ht.hip, XP = 6; mb_gen_cfg.precision 2 / 3): corrections for the fp16 rounding
of the weights (e2m1 of the operand VALUES against e2m1(W - fp16(W))) and of the LayerNorm outputs (e2m1 of their lo halves against e2m1(W)) as
24 KiB mini-tiles multiplied between the fp16 K-tiles.  Checked against fp64 on the DECODED 4-bit operands -- what the kernel is asked to compute --
for pair and plain tiles, one and two operand sets, every epilogue, the producers of the 4-bit operands (LayerNorm, GELU epilogue), and for
independence of timing (the scale dwords arrive by loads the compiler does not track)."""
import pytest
import torch

from hip_helpers import f4_block_exponent, f4_codes, f4_decode, f4_encode_rows, f4_scale_index, gemm_mini, w4_decode

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _weights(N, K, lib, lo: bool):
    from maskbit_amd import _lib
    W32 = torch.randn(N, K, device=DEV) * 0.03 * (0.5 + torch.rand(N, 1, device=DEV) * 2)      # rows of different scale
    w4 = torch.zeros(N, 2 * K, device=DEV, dtype=torch.uint8)
    wsb = torch.zeros(N * K // 128, device=DEV, dtype=torch.uint8)
    fn = lib.mb_w4lo_from_f32 if lo else lib.mb_w4_from_f32
    _lib.check(fn(W32.data_ptr(), N, K, w4.data_ptr(), wsb.data_ptr(), torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    return W32, w4, wsb, w4_decode(w4, wsb, N, K)


@pytest.mark.parametrize("epi,pairs,N,K,nlo", [(0, 2, 768, 1024, 1), (2, 3, 256, 2048, 1), (1, 2, 512, 1024, 1), (0, 2, 768, 1024, 2), (1, 3, 1024, 1024, 2),
                                               (2, 2, 1024, 4096, 1), (2, 3, 1024, 1024, 2), (2, 2, 1024, 4096, 2)])       # (round 6, precision 4: two sets in the residual GEMMs too)
def test_pair_gemm_with_mini_tile_passes(epi, pairs, N, K, nlo):
    """Pair tiles: out_c = f(A_c.W^T + sum_sets A4.W4^T + b), out_u = f(the same + A_delta.W^T) -- the corrections reach the unconditional rows through
    the shared conditional accumulator; class-token rows and the difference rows' own 4-bit data take no part."""
    from maskbit_amd import _lib
    lib = _lib.load()
    torch.manual_seed(epi + pairs + nlo)
    P = pairs * 257
    xc = torch.randn(P, K, device=DEV) * (0.2 + torch.rand(P, K // 64, device=DEV).repeat_interleave(64, 1) * 3)     # blocks of very different scale
    xu = xc + torch.randn(P, K, device=DEV) * 0.05
    A = torch.cat([xc.half(), (xu - xc).half()])
    W32, w4lo, wslo, wlo_dec = _weights(N, K, lib, lo=True)
    W = W32.half()
    x4, xs, x4_dec = f4_encode_rows(A[:P].double(), pairs)
    sets = [(x4, xs, w4lo, wslo)]
    corr = x4_dec @ wlo_dec.t()
    assert float(((x4_dec - A[:P].double()) ** 2)[x4_dec != 0].sum() / (A[:P].double() ** 2).sum()) < 0.05
    if nlo == 2:                                                   # activation-lo set: e2m1(x - fp16(x)) against e2m1(fp16(W))
        lo = (xc.double() - A[:P].double())
        _, w4v, wsv, wv_dec = _weights(N, K, lib, lo=False)        # (an unrelated weight: only the arithmetic is checked here)
        xl4, xls, xl_dec = f4_encode_rows(lo, pairs)
        sets.append((xl4, xls, w4v, wsv))
        corr = corr + xl_dec @ wv_dec.t()
    bias = torch.randn(N, device=DEV) * 0.1
    res = torch.randn(2 * P, N, device=DEV) if epi == 2 else None
    out32 = res.clone() if epi == 2 else None
    out16 = torch.full((2 * P, N), float("nan"), device=DEV, dtype=torch.float16) if epi != 2 else None
    # GELU epilogue with two sets = the engine's FFN-up at precision 4: also the e2m1 copies of the conditional outputs and of their fp16 lo halves
    f4o = epi == 1 and nlo == 2
    mk4 = lambda: (torch.zeros(2 * P, 2 * N, device=DEV, dtype=torch.uint8), torch.zeros((N // 64) * pairs * 256 + 256, device=DEV, dtype=torch.uint8))
    (out4, out4s), (out4l, out4ls) = (mk4(), mk4()) if f4o else ((None, None), (None, None))
    gemm_mini(lib, epi, A, W, bias, out32, out32, out16, P, True, N, K, sets, out4, out4s, 0, out4l, out4ls)
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
    if f4o:
        # the two e2m1 copies of the conditional GELU outputs h_c: values, and lo halves h_c - fp16(h_c) (the fp32 accumulation of fp16 products is within
        # ~1e-6 of the fp64 `want`, far below an fp16 ulp: the kernel's lo halves can be checked against want - fp16 row directly)
        rows = torch.arange(P)
        seq, tok = rows // 257, rows % 257
        keep = tok < 256
        hc = want[:P][keep.to(DEV)].cpu()
        for got4, gots, ref, tag in ((out4, out4s, hc, "values"), (out4l, out4ls, hc - out16[:P][keep.to(DEV)].double().cpu(), "lo halves")):
            sb = torch.stack([gots.cpu()[f4_scale_index(b, pairs, seq[keep], tok[keep])] for b in range(N // 64)], 1).double()
            dec = (f4_decode(got4[:P][keep.to(DEV)], N).reshape(-1, N // 64, 64) * (2.0 ** (sb - 127)).unsqueeze(-1)).reshape(-1, N)
            resid = float(((dec - ref) ** 2).sum() / (ref ** 2).sum())
            print(f"e2m1 copy of the GELU outputs' {tag}: residual energy {resid:.3f}")
            assert resid < (0.02 if tag == "values" else 0.06), tag
            assert int(got4[P:].count_nonzero()) == 0              # nothing for the difference rows
    # timing independence: repeat with the caches thrashed in between, bit for bit
    first = (out32 if out32 is not None else out16).clone()
    for _ in range(3):
        junk = torch.empty(96 << 20, device=DEV, dtype=torch.float32).normal_(); del junk
        if out32 is not None: out32.copy_(res)
        gemm_mini(lib, epi, A, W, bias, out32, out32, out16, P, True, N, K, sets, out4, out4s, 0, out4l, out4ls)
        torch.cuda.synchronize()
        assert torch.equal(out32 if out32 is not None else out16, first)
    if epi == 2 and nlo == 1:
        true_c = A[:P].double() @ W32.double().t() + bias.double() + res[:P].double()
        keep = (torch.arange(P, device=DEV) % 257) < 256
        e_corr = float((got[:P] - true_c)[keep].pow(2).mean().sqrt())
        e_plain = float((A[:P].double() @ W.double().t() + bias.double() + res[:P].double() - true_c)[keep].pow(2).mean().sqrt())
        print(f"rms error of the conditional rows vs fp32 weights: fp16 weights {e_plain:.3e}, with the correction mini-tiles {e_corr:.3e}")
        assert e_corr < e_plain / 3


@pytest.mark.parametrize("epi,nseq,N,K,SQ", [(0, 3, 768, 1024, 257), (1, 4, 1024, 1024, 257), (2, 2, 256, 2048, 257), (2, 5, 1024, 4096, 257),
                                             (0, 2, 768, 1024, 1025), (1, 1, 1024, 1024, 1025), (2, 1, 256, 2048, 1025), (2, 3, 1024, 1024, 1025), (2, 17, 1024, 1024, 1025)])
def test_plain_gemm_with_weight_correction_mini_tiles(epi, nseq, N, K, SQ):
    """Plain sequence tiles (the unguided forward, precision >= 2): both 128-row halves of every 256-token tile take the operand set; with the GELU epilogue
    the kernel also emits the e2m1 copy of its outputs + lane-ordered block scales for the next GEMM.  SQ = 1025 (the 512 x 512 models, round 5): four tiles
    per sequence, the class row computed and stored by the last one, 16 groups of 64 tokens in the scale arrays; the fp32 + residual cases also walk the
    quarter- / half-column tiles (few sequences) and whole tiles (17 sequences)."""
    from maskbit_amd import _lib
    lib = _lib.load()
    torch.manual_seed(epi + nseq)
    M = nseq * SQ
    G = (SQ - 1) // 64
    x = (torch.randn(M, K, device=DEV) * (0.3 + torch.rand(M, K // 64, device=DEV).repeat_interleave(64, 1) * 2)).half()
    W32, w4lo, wslo, wlo_dec = _weights(N, K, lib, lo=True)
    W = W32.half()
    x4, xs, x4_dec = f4_encode_rows(x.double(), nseq, seq_rows=SQ)
    bias = torch.randn(N, device=DEV) * 0.1
    res = torch.randn(M, N, device=DEV) if epi == 2 else None
    out32 = res.clone() if epi == 2 else None
    out16 = torch.full((M, N), float("nan"), device=DEV, dtype=torch.float16) if epi != 2 else None
    out4 = torch.zeros(M, 2 * N, device=DEV, dtype=torch.uint8) if epi == 1 else None
    out4s = torch.zeros((N // 64) * nseq * G * 64 + 256, device=DEV, dtype=torch.uint8) if epi == 1 else None
    gemm_mini(lib, epi, x, W, bias, out32, out32, out16, M, False, N, K, [(x4, xs, w4lo, wslo)], out4, out4s, seq_rows=0 if SQ == 257 else SQ)
    torch.cuda.synchronize()
    want = x.double() @ W.double().t() + x4_dec @ wlo_dec.t() + bias.double()
    if epi == 1:
        want = torch.nn.functional.gelu(want)
    if res is not None:
        want = want + res.double()
    got = (out32 if out32 is not None else out16).double()
    assert torch.isfinite(got).all()
    err = (got - want).abs()
    tol = 6e-5 if epi == 2 else 2e-3 * max(1.0, float(want.abs().max()))
    assert float(err.max()) < tol, (float(err.max()), int(err.argmax()) // N, int(err.argmax()) % N)
    if epi == 1:                                                   # the e2m1 copy of the GELU outputs: a valid quantisation of them, scales where the consumer reads them
        rows = torch.arange(M)
        seq, tok = rows // SQ, rows % SQ
        keep = tok < SQ - 1
        sb = torch.stack([out4s.cpu()[f4_scale_index(b, nseq, seq[keep], tok[keep], G)] for b in range(N // 64)], 1).double()      # [rows, blocks]
        dec = (f4_decode(out4[keep.to(DEV)], N).reshape(-1, N // 64, 64) * (2.0 ** (sb - 127)).unsqueeze(-1)).reshape(-1, N)
        ref = want[keep.to(DEV)].cpu()
        amax = ref.reshape(-1, N // 64, 64).abs().amax(-1)
        E = f4_block_exponent(amax.clamp(min=1e-30))
        assert float((sb != (E - 2).clamp(min=0).double()).double().mean()) < 2e-3          # (fp32 vs fp64 block maxima may straddle a binade)
        assert float(((dec - ref) ** 2).sum() / (ref ** 2).sum()) < 0.02


@pytest.mark.parametrize("epi,nseq,N,K", [(0, 3, 768, 1024), (1, 2, 1024, 1024)])
def test_plain_gemm_split_activations_with_weight_correction_mini_tiles(epi, nseq, N, K):
    """The plain forward's FFN-up of the late layers at precision >= 2 (mb_gemm_mini_split): the fp16 sweep runs over the hi
    AND the lo halves of the activations (K-tiles doubled) while the mini-tiles -- one per two fp16 K-tiles then, both row halves -- add the weight
    correction of the K columns: out = (x_hi + x_lo) . W^T + e2m1(x) . e2m1(W32 - W)^T + bias."""
    import ctypes as C
    from maskbit_amd import _lib
    lib = _lib.load()
    torch.manual_seed(10 * epi + nseq)
    M = nseq * 257
    x32 = torch.randn(M, K, device=DEV) * (0.3 + torch.rand(M, K // 64, device=DEV).repeat_interleave(64, 1) * 2)
    hi = x32.half()
    lo = (x32 - hi.float()).half()
    W32, w4lo, wslo, wlo_dec = _weights(N, K, lib, lo=True)
    W = W32.half()
    x4, xs, x4_dec = f4_encode_rows(hi.double(), nseq)
    bias = torch.randn(N, device=DEV) * 0.1
    out16 = torch.full((M, N), float("nan"), device=DEV, dtype=torch.float16)
    out4 = torch.zeros(M, 2 * N, device=DEV, dtype=torch.uint8) if epi == 1 else None
    out4s = torch.zeros((N // 64) * nseq * 256 + 256, device=DEV, dtype=torch.uint8) if epi == 1 else None
    ptr = lambda t: t.data_ptr() if t is not None else None
    arr = (C.c_void_p * 4)(x4.data_ptr(), xs.data_ptr(), w4lo.data_ptr(), wslo.data_ptr())
    _lib.check(lib.mb_gemm_mini_split(epi, hi.data_ptr(), lo.data_ptr(), W.data_ptr(), bias.data_ptr(), out16.data_ptr(), ptr(out4), ptr(out4s), M, N, K, arr,
                                      torch.cuda.current_stream().cuda_stream), "mb_gemm_mini_split")
    torch.cuda.synchronize()
    want = (hi.double() + lo.double()) @ W.double().t() + x4_dec @ wlo_dec.t() + bias.double()
    if epi == 1:
        want = torch.nn.functional.gelu(want)
    got = out16.double()
    assert torch.isfinite(got).all()
    err = (got - want).abs()
    assert float(err.max()) < 2e-3 * max(1.0, float(want.abs().max())), (float(err.max()), int(err.argmax()) // N, int(err.argmax()) % N)
    # against the fp32 operands: closer than either half measure (hi + lo activations with fp16 weights; single fp16 activations with the correction)
    true = x32.double() @ W32.double().t() + bias.double()
    if epi == 1:
        true = torch.nn.functional.gelu(true)
    keep = (torch.arange(M, device=DEV) % 257) < 256
    rms = lambda t: float((t - true)[keep].pow(2).mean().sqrt())
    f = (lambda t: torch.nn.functional.gelu(t)) if epi == 1 else (lambda t: t)
    e_both = rms(got)
    e_act = rms(f((hi.double() + lo.double()) @ W.double().t() + bias.double()).half().double())
    e_w = rms(f(hi.double() @ W.double().t() + x4_dec @ wlo_dec.t() + bias.double()).half().double())
    print(f"rms error vs fp32 operands: hi + lo activations alone {e_act:.3e}, weight correction alone {e_w:.3e}, both {e_both:.3e}")
    assert e_both <= 1.02 * min(e_act, e_w)


@pytest.mark.parametrize("d,nseq", [(1024, 3), (768, 2)])
def test_layernorm_writes_e2m1_values_and_lo_halves(d, nseq):
    """mb_layernorm_f4 (the LayerNorm kernels' producer path): e2m1 of the normalised rows and of their fp16 lo halves, one scale per 64 columns
    chosen from the block's largest element (3 < max <= 6), bytes and lane-ordered scale bytes exactly as the rule of mb_common.h prescribes;
    class-token rows are left alone."""
    from maskbit_amd import _lib
    lib = _lib.load()
    torch.manual_seed(d)
    M = nseq * 257
    y = torch.randn(M, d, device=DEV) * 3.0
    y[:, 5] += 40.0                                                # a massive-activation channel: costs the resolution of ITS block only
    gam, bet = torch.rand(d, device=DEV) + 0.5, torch.randn(d, device=DEV) * 0.2
    x32 = torch.empty(M, d, device=DEV)
    xh = torch.empty(M, d, device=DEV, dtype=torch.float16)
    x4 = torch.full((M, 2 * d), 0xAB, device=DEV, dtype=torch.uint8)
    xl4 = torch.full((M, 2 * d), 0xCD, device=DEV, dtype=torch.uint8)
    ns = (d // 64) * nseq * 256 + 256
    x4s = torch.full((ns,), 0xEE, device=DEV, dtype=torch.uint8)
    xl4s = torch.full((ns,), 0xEE, device=DEV, dtype=torch.uint8)
    _lib.check(lib.mb_layernorm_f4(y.data_ptr(), gam.data_ptr(), bet.data_ptr(), 1e-12, x32.data_ptr(), xh.data_ptr(), x4.data_ptr(), x4s.data_ptr(),
                                   xl4.data_ptr(), xl4s.data_ptr(), M, d, torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert torch.equal(xh, x32.half())
    rows = torch.arange(M)
    seq, tok = rows // 257, rows % 257
    keep = tok < 256
    for got4, gots, v in ((x4, x4s, x32.double().cpu()), (xl4, xl4s, (x32 - xh.float()).double().cpu())):
        blocks = v.reshape(M, d // 64, 64)
        E = f4_block_exponent(blocks.abs().amax(-1).clamp(min=1e-30))
        codes = f4_codes(blocks * (2.0 ** (129 - E).double()).unsqueeze(-1)).reshape(M, d)
        want = (codes[:, 0::2] | (codes[:, 1::2] << 4)).to(torch.uint8)
        g4 = got4.cpu()
        assert torch.equal(g4[keep][:, : d // 2], want[keep]), f"{int((g4[keep][:, : d // 2] != want[keep]).sum())} e2m1 bytes differ"
        assert bool((g4[~keep][:, : d // 2] == g4[~keep][0, 0]).all())                  # class-token rows untouched
        gs = gots.cpu()
        for b in range(d // 64):
            assert torch.equal(gs[f4_scale_index(b, nseq, seq[keep], tok[keep])].to(torch.int64), (E[keep, b] - 2).clamp(min=0).to(torch.int64))
    # the outlier channel sits in block 0: the other blocks keep their own resolution
    dec_err = lambda blk: float(((f4_decode(x4.cpu()[keep], d).reshape(-1, d // 64, 64)[:, blk] *
                                  (2.0 ** (x4s.cpu()[f4_scale_index(blk, nseq, seq[keep], tok[keep])].double() - 127)).unsqueeze(-1)
                                  - x32.double().cpu()[keep].reshape(-1, d // 64, 64)[:, blk]) ** 2).mean())
    assert dec_err(3) < 0.1 * dec_err(0)
