"""GPU tests of the trunk GEMM family (both kernels, every tile height, every epilogue) against a
plain PyTorch fp32 matmul of the same fp16 operands.  Tolerance: fp32 accumulation-order noise plus
(for fp16 outputs) one fp16 rounding of the result."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _run(epi, A, W, bias, res, variant, period=0):
    from maskbit_amd import _lib
    lib = _lib.load()
    M, K = A.shape
    N = W.shape[0]
    rows = M if epi != 4 else (M // period) * (period - 1)
    out32 = torch.full((rows, N), float("nan"), device=DEV, dtype=torch.float32) if epi in (2, 3, 4) else None
    out16 = torch.full((M, N), float("nan"), device=DEV, dtype=torch.float16) if epi in (0, 1) else None
    _lib.check(lib.mb_gemm(epi, A.data_ptr(), W.data_ptr(), bias.data_ptr(), res.data_ptr() if res is not None else None,
                           out32.data_ptr() if out32 is not None else None, out16.data_ptr() if out16 is not None else None,
                           M, N, K, period, variant, torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    return out32 if out32 is not None else out16


def _ref(epi, A, W, bias, res, period=0):
    y = A.float() @ W.float().t() + bias
    if epi == 2:
        y = y + res
    if epi in (1, 3):
        y = torch.nn.functional.gelu(y)
    if epi == 4:
        y = y.reshape(A.shape[0] // period, period, -1)[:, :period - 1].reshape(-1, y.shape[-1])
    return y


@pytest.mark.parametrize("variant", [-1, 6, 8, 0])
@pytest.mark.parametrize("epi,N,K", [(0, 768, 256), (1, 512, 128), (2, 256, 512), (3, 256, 64)])
def test_gemm_variants_match_torch(variant, epi, N, K):
    torch.manual_seed(variant * 10 + epi)
    M = 5 * 257 + 3                                         # ragged: not a multiple of any tile height
    A = torch.randn(M, K, device=DEV).half()
    W = (torch.randn(N, K, device=DEV) * 0.1).half()        # asymmetric operands: catches transposed fragments
    bias = torch.randn(N, device=DEV)
    res = torch.randn(M, N, device=DEV) if epi == 2 else None
    out = _run(epi, A, W, bias, res, variant)
    ref = _ref(epi, A, W, bias, res)
    assert not torch.isnan(out.float()).any()               # every output element written exactly once
    tol = 2e-3 * float(ref.abs().max()) if epi in (0, 1) else 2e-4 * float(ref.abs().max())
    assert float((out.float() - ref).abs().max()) < tol


@pytest.mark.parametrize("epi,N,K", [(0, 768, 256), (1, 512, 128), (2, 256, 512), (3, 256, 64)])
def test_gemm_sequence_aligned_tiles(epi, N, K):
    """M = nb*257: one tile per sequence (256 token rows + the class-token row as a 17th one-row m-tile)."""
    torch.manual_seed(100 + epi)
    M = 6 * 257
    A = torch.randn(M, K, device=DEV).half()
    W = (torch.randn(N, K, device=DEV) * 0.1).half()
    bias = torch.randn(N, device=DEV)
    res = torch.randn(M, N, device=DEV) if epi == 2 else None
    ref = _ref(epi, A, W, bias, res)
    tol = 2e-3 * float(ref.abs().max()) if epi in (0, 1) else 2e-4 * float(ref.abs().max())
    for variant in (257, 0):
        out = _run(epi, A, W, bias, res, variant)
        assert not torch.isnan(out.float()).any()
        err = (out.float() - ref).abs()
        assert float(err.max()) < tol, f"variant {variant}: worst row {int(err.max(1).values.argmax())} (class rows are 256 mod 257)"


def test_gemm_identity_detects_layout_bugs():
    """A = I (padded) with an asymmetric W returns W^T rows exactly: row/col swaps or fragment permutations show up."""
    K = N = 256
    M = 1024
    A = torch.zeros(M, K, device=DEV, dtype=torch.float16)
    A[torch.arange(K), torch.arange(K)] = 1.0
    A[K:2 * K] = A[:K] * 2
    W = (torch.arange(N * K, device=DEV, dtype=torch.float32).reshape(N, K) % 251 - 125).half()
    bias = torch.zeros(N, device=DEV)
    for variant in (-1, 6, 8):
        out = _run(3 if False else 2, A, W, bias, torch.zeros(M, N, device=DEV), variant)
        assert torch.equal(out[:K], W.float().t()) and torch.equal(out[K:2 * K], 2 * W.float().t())
        assert float(out[2 * K:].abs().max()) == 0.0


def test_gemm_logits_epilogue_drops_class_rows():
    torch.manual_seed(3)
    period, nb, K, N = 257, 3, 128, 128
    A = torch.randn(nb * period, K, device=DEV).half()
    W = (torch.randn(N, K, device=DEV) * 0.1).half()
    bias = torch.randn(N, device=DEV)
    out = _run(4, A, W, bias, None, 0, period)
    ref = _ref(4, A, W, bias, None, period)
    assert out.shape == (nb * 256, N) and float((out - ref).abs().max()) < 2e-4 * float(ref.abs().max())


@pytest.mark.parametrize("variant", [-1, 8, 0])
@pytest.mark.parametrize("epi,M,N,K", [(2, 1028, 512, 1024), (0, 771, 256, 128), (4, 1028, 128, 256)])
def test_split_weight_gemm(variant, epi, M, N, K):
    """fp16x2 weights (mb_split_weights + mb_gemm_ex): the product must track the fp32 weights, i.e. be far closer to an
    fp64 reference than the same GEMM with weights rounded once to fp16, and the repack must be exact to ~2^-22."""
    from maskbit_amd import _lib
    lib = _lib.load()
    torch.manual_seed(epi * 7 + variant)
    period = 257 if epi == 4 else 0
    A = torch.randn(M, K, device=DEV).half()
    W = torch.randn(N, K, device=DEV) * 0.02
    W[0, 0] = 0.37                                                 # sets the power-of-two scale; most weights are much smaller
    bias = torch.randn(N, device=DEV) * 0.1
    res = torch.randn(M, N, device=DEV) if epi == 2 else None
    W2 = torch.empty(N, 2 * K, device=DEV, dtype=torch.float16)
    scale = torch.zeros(1, device=DEV)
    tmp = torch.zeros(1, device=DEV, dtype=torch.int32)
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.mb_split_weights(W.data_ptr(), N, K, W2.data_ptr(), scale.data_ptr(), tmp.data_ptr(), st))
    torch.cuda.synchronize()
    s = float(scale)
    assert s == 2.0 ** -16                                          # 0.37 * 2^16 = 24248 in [2^14, 2^15)
    back = (W2[:, :K].double() + W2[:, K:].double()) * s
    assert float((back - W.double()).abs().max()) <= 2.0 ** -22 * 0.37
    rows = M if epi != 4 else (M // period) * (period - 1)
    out32 = torch.full((rows, N), float("nan"), device=DEV) if epi in (2, 4) else None
    out16 = torch.full((M, N), float("nan"), device=DEV, dtype=torch.float16) if epi == 0 else None
    _lib.check(lib.mb_gemm_ex(epi, A.data_ptr(), W2.data_ptr(), bias.data_ptr(), res.data_ptr() if res is not None else None,
                              out32.data_ptr() if out32 is not None else None, out16.data_ptr() if out16 is not None else None,
                              M, N, 2 * K, K, scale.data_ptr(), None, None, None, period, variant, st))
    torch.cuda.synchronize()
    ref = A.double() @ W.double().t() + bias.double()
    if res is not None:
        ref = ref + res.double()
    if epi == 4:
        ref = ref.reshape(M // period, period, -1)[:, :period - 1].reshape(-1, N)
    got = (out32 if out32 is not None else out16).double()
    assert torch.isfinite(got).all()
    err = float((got - ref).abs().max())
    if epi == 0:
        assert err < 4e-3                                           # one fp16 rounding of the result
    else:
        single = _run(epi, A, W.half(), bias, res, variant, period).double()
        err_single = float((single - ref).abs().max())
        print(f"max err split {err:.2e}  single-fp16 weights {err_single:.2e}")
        assert err < 2e-5 and err < err_single / 8


@pytest.mark.parametrize("variant", [-1, 6, 8, 0])
@pytest.mark.parametrize("M,N,K", [(1028, 512, 256), (771, 256, 128), (600, 768, 192)])
def test_layernorm_residual_epilogue(variant, M, N, K):
    """The fp32+residual GEMM can take the PRE-LayerNorm rows plus {mean, rstd} and re-derive the normalised residual in its
    epilogue, in place.  It must equal, bit for bit, the plain-residual GEMM fed with the fp32 rows mb_layernorm stores."""
    from maskbit_amd import _lib
    lib = _lib.load()
    torch.manual_seed(M + variant)
    st = torch.cuda.current_stream().cuda_stream
    A = torch.randn(M, K, device=DEV).half()
    W = (torch.randn(N, K, device=DEV) * 0.05).half()
    bias = torch.randn(N, device=DEV) * 0.1
    y = torch.randn(M, N, device=DEV) * 1.7 + 0.3
    g = torch.rand(N, device=DEV) + 0.5
    b = torch.randn(N, device=DEV) * 0.2
    x32 = torch.empty_like(y); x16 = torch.empty(M, N, device=DEV, dtype=torch.float16); stats = torch.empty(M, 2, device=DEV)
    _lib.check(lib.mb_layernorm(y.data_ptr(), g.data_ptr(), b.data_ptr(), 1e-12, x32.data_ptr(), x16.data_ptr(), None, stats.data_ptr(), M, N, st))
    torch.cuda.synchronize()
    ref_ln = torch.nn.functional.layer_norm(y, (N,), g, b, 1e-12)
    assert float((x32 - ref_ln).abs().max()) < 2e-5
    assert float((stats[:, 0] - y.mean(1)).abs().max()) < 1e-5
    assert torch.equal(x16, x32.half())
    plain = _run(2, A, W, bias, x32, variant)
    buf = y.clone()                                                   # in place: residual == out
    _lib.check(lib.mb_gemm_ex(2, A.data_ptr(), W.data_ptr(), bias.data_ptr(), buf.data_ptr(), buf.data_ptr(), None, M, N, K, 0, None,
                              stats.data_ptr(), g.data_ptr(), b.data_ptr(), 0, variant, st))
    torch.cuda.synchronize()
    assert torch.equal(buf, plain)
    ref = A.float() @ W.float().t() + bias + ref_ln
    assert float((buf - ref).abs().max()) < 2e-3


@pytest.mark.parametrize("epi,M,N,K", [(0, 1300, 512, 256), (1, 1024, 256, 128), (2, 771, 768, 192), (3, 600, 256, 64 * 5)])
def test_four_wave_variant_matches_half_tile_kernel(epi, M, N, K):
    """The experimental 4-wave kernel (variant 4) accumulates in the same order: bit-identical to the production kernel."""
    torch.manual_seed(epi)
    A = torch.randn(M, K, device=DEV).half()
    W = (torch.randn(N, K, device=DEV) * 0.05).half()
    bias = torch.randn(N, device=DEV) * 0.1
    res = torch.randn(M, N, device=DEV) if epi == 2 else None
    assert torch.equal(_run(epi, A, W, bias, res, 4), _run(epi, A, W, bias, res, 8))


@pytest.mark.parametrize("variant", [-1, 8, 257, 0])
@pytest.mark.parametrize("epi,M,N,K", [(2, 1028, 512, 1024), (0, 771, 256, 128), (1, 1028, 1024, 256), (2, 200, 128, 192)])
def test_split_activation_gemm(variant, epi, M, N, K):
    """fp16 hi+lo activation pairs (mb_gen_cfg.act_split): mb_layernorm writes x_hi and x_lo = fp16(x - x_hi); the GEMM over the pair
    sweeps W twice (K-tiles 0..K/64-1 take x_hi, the rest x_lo) and must track the fp32 LayerNorm rows far better than x_hi alone."""
    from maskbit_amd import _lib
    lib = _lib.load()
    if variant == 257 and M % 257:
        pytest.skip("sequence-aligned tiles need M % 257 == 0")
    if variant in (8, 257) and (M < 512 or N % 256 or 2 * K < 128):
        pytest.skip("half-tile kernel shape limits")
    torch.manual_seed(epi * 11 + (variant & 7))
    y = torch.randn(M, K, device=DEV) * 3.0
    g = torch.rand(K, device=DEV) + 0.5
    b = torch.randn(K, device=DEV) * 0.2
    st = torch.cuda.current_stream().cuda_stream
    x32 = torch.empty(M, K, device=DEV)
    xh = torch.empty(M, K, device=DEV, dtype=torch.float16)
    xl = torch.empty(M, K, device=DEV, dtype=torch.float16)
    _lib.check(lib.mb_layernorm(y.data_ptr(), g.data_ptr(), b.data_ptr(), 1e-12, x32.data_ptr(), xh.data_ptr(), xl.data_ptr(), None, M, K, st))
    torch.cuda.synchronize()
    assert torch.equal(xh, x32.half()) and torch.equal(xl, (x32 - xh.float()).half())           # exactly the hi / lo halves of the fp32 rows
    assert float((xh.double() + xl.double() - x32.double()).abs().max()) < 2.0 ** -20
    W = (torch.randn(N, K, device=DEV) * 0.05).half()
    bias = torch.randn(N, device=DEV) * 0.1
    res = torch.randn(M, N, device=DEV) if epi == 2 else None
    out32 = torch.full((M, N), float("nan"), device=DEV) if epi == 2 else None
    out16 = torch.full((M, N), float("nan"), device=DEV, dtype=torch.float16) if epi != 2 else None
    _lib.check(lib.mb_gemm_act_split(epi, xh.data_ptr(), xl.data_ptr(), W.data_ptr(), bias.data_ptr(), res.data_ptr() if res is not None else None,
                                     out32.data_ptr() if out32 is not None else None, out16.data_ptr() if out16 is not None else None,
                                     M, N, K, variant, st))
    torch.cuda.synchronize()
    ref = x32.double() @ W.double().t() + bias.double()
    if epi == 1:
        ref = torch.nn.functional.gelu(ref)
    if res is not None:
        ref = ref + res.double()
    got = (out32 if out32 is not None else out16).double()
    assert torch.isfinite(got).all()
    err = float((got - ref).abs().max())
    if epi == 2:
        single = _run(epi, xh, W, bias, res, variant if variant != 257 else 0, 0).double()
        err_single = float((single - ref).abs().max())
        print(f"max err vs fp32 rows: hi+lo {err:.2e}, hi only {err_single:.2e}")
        assert err < 3e-5 and err < err_single / 8
    else:
        assert err < 2e-3 * max(1.0, float(ref.abs().max()))       # one fp16 rounding of the result


@pytest.mark.parametrize("variant", [8, 257, 0])
@pytest.mark.parametrize("epi,M,N,K", [(2, 1028, 512, 1024), (0, 771, 256, 128), (1, 1028, 1024, 256), (2, 257, 256, 384)])
def test_f8_lo_pass_gemm(variant, epi, M, N, K):
    """The e4m3 lo pass of a split-activation GEMM (mb_gen_cfg.act_split == 3): K-tiles of the fp16 pair (x_hi, W), then K/128 e4m3 K-tiles of
    (e4m3(x_lo * 2^12), e4m3(W * 2^e)) on v_mfma_scale_f32_16x16x128_f8f6f4, whose E8M0 scales undo the two powers of two.  Checked
    (a) against the exact value of what the kernel is asked to compute (decoded e4m3 operands, fp64) and (b) against the fp32 rows:
    the result must be far closer to them than the hi halves alone."""
    from maskbit_amd import _lib
    lib = _lib.load()
    if variant == 257 and M % 257:
        pytest.skip("sequence-aligned tiles need M % 257 == 0")
    torch.manual_seed(epi * 13 + (variant & 7))
    x32 = torch.randn(M, K, device=DEV) * 1.5
    xh = x32.half()
    lo = x32 - xh.float()
    a8 = torch.zeros(M, 2 * K, device=DEV, dtype=torch.uint8)
    a8[:, :K] = (lo * 2.0 ** 12).to(torch.float8_e4m3fn).view(torch.uint8)
    W32 = torch.randn(N, K, device=DEV) * 0.05
    W = W32.half()
    e = 10                                                       # |W| < 0.25 -> |W * 2^10| < 256
    w8 = torch.zeros(N, 2 * K, device=DEV, dtype=torch.uint8)
    w8[:, :K] = (W.float() * 2.0 ** e).to(torch.float8_e4m3fn).view(torch.uint8)
    wexp = torch.tensor([e], device=DEV, dtype=torch.int32)
    bias = torch.randn(N, device=DEV) * 0.1
    res = torch.randn(M, N, device=DEV) if epi == 2 else None
    out32 = torch.full((M, N), float("nan"), device=DEV) if epi == 2 else None
    out16 = torch.full((M, N), float("nan"), device=DEV, dtype=torch.float16) if epi != 2 else None
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.mb_gemm_f8lo(epi, xh.data_ptr(), a8.data_ptr(), W.data_ptr(), w8.data_ptr(), wexp.data_ptr(), bias.data_ptr(),
                                res.data_ptr() if res is not None else None, out32.data_ptr() if out32 is not None else None,
                                out16.data_ptr() if out16 is not None else None, M, N, K, variant, st))
    torch.cuda.synchronize()
    lo_dec = a8[:, :K].view(torch.float8_e4m3fn).double() / 2.0 ** 12
    w_dec = w8[:, :K].view(torch.float8_e4m3fn).double() / 2.0 ** e
    asked = xh.double() @ W.double().t() + lo_dec @ w_dec.t() + bias.double()
    true = x32.double() @ W.double().t() + bias.double()
    if epi == 1:
        asked, true = torch.nn.functional.gelu(asked), torch.nn.functional.gelu(true)
    if res is not None:
        asked, true = asked + res.double(), true + res.double()
    got = (out32 if out32 is not None else out16).double()
    assert torch.isfinite(got).all()
    if epi == 2:
        hi_only = (xh.double() @ W.double().t() + bias.double() + res.double())
        e_asked, e_true, e_hi = (float((got - r).abs().max()) for r in (asked, true, hi_only))
        print(f"max err vs the asked value {e_asked:.2e}, vs the fp32 rows {e_true:.2e}; hi halves alone are {float((hi_only - true).abs().max()):.2e} away")
        assert e_asked < 3e-5 and e_true < float((hi_only - true).abs().max()) / 8
    else:
        assert float((got - asked).abs().max()) < 2e-3 * max(1.0, float(asked.abs().max()))


# ---- MX-fp4 lo pass (mb_gen_cfg.act_split == 4) ------------------------------------------------------------------------------------
_F4 = torch.tensor([0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0, -0.0, -0.5, -1.0, -1.5, -2.0, -3.0, -4.0, -6.0], dtype=torch.float64)


def _f4_decode(buf: torch.Tensor, K: int) -> torch.Tensor:
    """uint8 [R, >= K/2] (element 2j = low nibble of byte j) -> float64 [R, K] of e2m1 values."""
    b = buf[:, : K // 2].cpu().to(torch.int64)
    codes = torch.stack([b & 15, b >> 4], dim=-1).reshape(b.shape[0], K)
    return _F4[codes]


def _f4_codes(v: torch.Tensor) -> torch.Tensor:
    """float64 scaled values -> e2m1 codes 0..15 the way mb_common.h fp4_code rounds (nearest, ties to even on the local grid, saturate at 6)."""
    a = v.abs().clamp(max=6.0)
    k = torch.floor(torch.log2(a.clamp(min=1.0))).clamp(0, 2)
    r = torch.round(a / 2.0 ** (k - 1))
    return (r + 2 * k).to(torch.int64) | ((v < 0).to(torch.int64) << 3)


@pytest.mark.parametrize("variant", [8, 257, 0])
@pytest.mark.parametrize("epi,M,N,K", [(2, 1028, 512, 1024), (0, 771, 256, 256), (1, 1028, 1024, 1024), (2, 257, 256, 768)])
def test_f4_lo_pass_gemm(variant, epi, M, N, K):
    """The MX-fp4 lo pass: K/64 fp16 K-tiles of (x_hi, W), then K/256 K-tiles of 256 e2m1 values per row of (e2m1(x_lo * 2^s_m), e2m1(W * 2^r_n))
    on v_mfma_scale_f32_16x16x128_f8f6f4 (cbsz = blgp = 4) whose per-lane E8M0 scales undo 2^s_m and 2^r_n.  The operands come from the engine's
    own producers (mb_layernorm_f4, mb_w4_from_f32), which are checked first: LayerNorm bytes / scale bytes exactly against the rule of
    mb_common.h, weight codes as a valid quantisation of fp16(W) within half a grid step.  The GEMM is checked (a) against the exact value of
    what it is asked to compute (decoded 4-bit operands, fp64) and (b) against the fp32 rows: far closer than the hi halves alone."""
    from maskbit_amd import _lib
    lib = _lib.load()
    if variant == 257 and M % 257:
        pytest.skip("sequence-aligned tiles need M % 257 == 0")
    torch.manual_seed(epi * 17 + (variant & 7))
    st = torch.cuda.current_stream().cuda_stream
    d = K
    if d not in (768, 1024):                             # the fp4 LayerNorm output exists for the engine's widths; build the operands by hand otherwise
        x32 = torch.randn(M, K, device=DEV) * 1.5
        xh = x32.half()
        lo = (x32 - xh.float()).double()
        maxlo = lo.abs().amax(1, keepdim=True)
        e = torch.floor(torch.log2(maxlo)).to(torch.int64) + 127
        sbyte = (e - 2).clamp(min=0).to(torch.uint8).reshape(M)
        codes = _f4_codes(lo * 2.0 ** (129 - e).double())
        x4 = torch.zeros(M, 2 * K, device=DEV, dtype=torch.uint8)
        x4[:, : K // 2] = (codes[:, 0::2] | (codes[:, 1::2] << 4)).to(torch.uint8)
    else:
        y = torch.randn(M, K, device=DEV) * 3.0
        gam = torch.rand(K, device=DEV) + 0.5
        bet = torch.randn(K, device=DEV) * 0.2
        x32 = torch.empty(M, K, device=DEV)
        xh = torch.empty(M, K, device=DEV, dtype=torch.float16)
        x4 = torch.zeros(M, 2 * K, device=DEV, dtype=torch.uint8)
        sbyte = torch.zeros(M, device=DEV, dtype=torch.uint8)
        _lib.check(lib.mb_layernorm_f4(y.data_ptr(), gam.data_ptr(), bet.data_ptr(), 1e-12, x32.data_ptr(), xh.data_ptr(), x4.data_ptr(), sbyte.data_ptr(), M, K, st))
        torch.cuda.synchronize()
        assert torch.equal(xh, x32.half())
        lo = (x32 - xh.float()).double()
        maxlo = lo.abs().amax(1, keepdim=True)
        e = torch.floor(torch.log2(maxlo)).to(torch.int64) + 127
        assert torch.equal(sbyte.to(torch.int64), (e - 2).reshape(M))
        codes = _f4_codes(lo * 2.0 ** (129 - e).double())
        want = (codes[:, 0::2] | (codes[:, 1::2] << 4)).to(torch.uint8)
        assert torch.equal(x4[:, : K // 2], want), f"{int((x4[:, : K // 2] != want).sum())} LayerNorm fp4 bytes differ"
    W32 = torch.randn(N, K, device=DEV) * 0.05 * (0.5 + torch.rand(N, 1, device=DEV) * 2)      # rows of different scale
    W = W32.half()
    w4 = torch.zeros(N, 2 * K, device=DEV, dtype=torch.uint8)
    wsb = torch.zeros(N, device=DEV, dtype=torch.uint8)
    _lib.check(lib.mb_w4_from_f32(W32.data_ptr(), N, K, w4.data_ptr(), wsb.data_ptr(), st))
    torch.cuda.synchronize()
    n = torch.arange(N, device=DEV)
    wscale_row = wsb[((n >> 6) * 16 + (n & 15)) * 4 + ((n >> 4) & 3)].to(torch.float64)        # un-permute the lane order
    w_dec = _f4_decode(w4, K).to(DEV) * (2.0 ** (wscale_row - 127)).reshape(N, 1)
    rel = float(((w_dec - W.double()) ** 2).sum() / (W.double() ** 2).sum())
    assert rel < 0.03, f"fp4 weight copy: relative squared error {rel:.3f}"
    lo_dec = _f4_decode(x4, K).to(DEV) * (2.0 ** (sbyte.to(torch.float64) - 127)).reshape(M, 1)
    rel_lo = float(((lo_dec - lo) ** 2).sum() / (lo ** 2).sum())
    assert rel_lo < 0.05, f"fp4 lo halves: residual variance ratio {rel_lo:.3f}"
    bias = torch.randn(N, device=DEV) * 0.1
    res = torch.randn(M, N, device=DEV) if epi == 2 else None
    out32 = torch.full((M, N), float("nan"), device=DEV) if epi == 2 else None
    out16 = torch.full((M, N), float("nan"), device=DEV, dtype=torch.float16) if epi != 2 else None
    _lib.check(lib.mb_gemm_f4lo(epi, xh.data_ptr(), x4.data_ptr(), sbyte.data_ptr(), W.data_ptr(), w4.data_ptr(), wsb.data_ptr(), bias.data_ptr(),
                                res.data_ptr() if res is not None else None, out32.data_ptr() if out32 is not None else None,
                                out16.data_ptr() if out16 is not None else None, M, N, K, variant, st))
    torch.cuda.synchronize()
    asked = xh.double() @ W.double().t() + lo_dec @ w_dec.t() + bias.double()
    true = x32.double() @ W.double().t() + bias.double()
    if epi == 1:
        asked, true = torch.nn.functional.gelu(asked), torch.nn.functional.gelu(true)
    if res is not None:
        asked, true = asked + res.double(), true + res.double()
    got = (out32 if out32 is not None else out16).double()
    assert torch.isfinite(got).all()
    if epi == 2:
        hi_only = (xh.double() @ W.double().t() + bias.double() + res.double())
        e_asked, e_true, e_hi = float((got - asked).abs().max()), float((got - true).abs().max()), float((hi_only - true).abs().max())
        rms_true, rms_hi = float((got - true).pow(2).mean().sqrt()), float((hi_only - true).pow(2).mean().sqrt())
        print(f"max err vs the asked value {e_asked:.2e}, vs the fp32 rows {e_true:.2e} (rms {rms_true:.2e}); hi halves alone: max {e_hi:.2e}, rms {rms_hi:.2e}")
        assert e_asked < 3e-5 and rms_true < rms_hi / 4
    else:
        assert float((got - asked).abs().max()) < 2e-3 * max(1.0, float(asked.abs().max()))


@pytest.mark.parametrize("M,N,K", [(1028, 512, 1024), (771, 1024, 1024)])
def test_f4_weight_correction_pass(M, N, K):
    """The weight-correction use of the MX-fp4 pass (mb_gen_cfg.cfg_pair == 2): A4 = e2m1 of the activation VALUES, W4 = e2m1 of the weight's fp16
    rounding error (mb_w4lo_from_f32).  x.W16^T + x4.Wlo4^T must be several times closer to x.W32^T than x.W16^T alone."""
    from maskbit_amd import _lib
    lib = _lib.load()
    torch.manual_seed(M)
    st = torch.cuda.current_stream().cuda_stream
    x = (torch.randn(M, K, device=DEV) * 1.2).half()
    W32 = torch.randn(N, K, device=DEV) * 0.02
    W = W32.half()
    xv = x.double()
    amax = xv.abs().amax(1, keepdim=True)
    e = torch.floor(torch.log2(amax)).to(torch.int64) + 127
    sbyte = (e - 2).clamp(min=0).to(torch.uint8).reshape(M)
    codes = _f4_codes(xv * 2.0 ** (129 - e).double())
    x4 = torch.zeros(M, 2 * K, device=DEV, dtype=torch.uint8)
    x4[:, : K // 2] = (codes[:, 0::2] | (codes[:, 1::2] << 4)).to(torch.uint8)
    w4 = torch.zeros(N, 2 * K, device=DEV, dtype=torch.uint8)
    wsb = torch.zeros(N, device=DEV, dtype=torch.uint8)
    _lib.check(lib.mb_w4lo_from_f32(W32.data_ptr(), N, K, w4.data_ptr(), wsb.data_ptr(), st))
    torch.cuda.synchronize()
    n = torch.arange(N, device=DEV)
    wscale_row = wsb[((n >> 6) * 16 + (n & 15)) * 4 + ((n >> 4) & 3)].to(torch.float64)
    wlo_dec = _f4_decode(w4, K).to(DEV) * (2.0 ** (wscale_row - 127)).reshape(N, 1)
    wlo = (W32.double() - W.double())
    rel = float(((wlo_dec - wlo) ** 2).sum() / (wlo ** 2).sum())
    assert rel < 0.05, f"fp4 copy of the weight rounding error: residual variance ratio {rel:.3f}"
    bias = torch.zeros(N, device=DEV)
    res = torch.zeros(M, N, device=DEV)
    out = torch.full((M, N), float("nan"), device=DEV)
    _lib.check(lib.mb_gemm_f4lo(2, x.data_ptr(), x4.data_ptr(), sbyte.data_ptr(), W.data_ptr(), w4.data_ptr(), wsb.data_ptr(), bias.data_ptr(),
                                res.data_ptr(), out.data_ptr(), None, M, N, K, 0, st))
    torch.cuda.synchronize()
    true = xv @ W32.double().t()
    plain = xv @ W.double().t()
    e_corr, e_plain = float((out.double() - true).pow(2).mean().sqrt()), float((plain - true).pow(2).mean().sqrt())
    print(f"rms error vs fp32 weights: fp16 weights {e_plain:.3e}, with the fp4 correction pass {e_corr:.3e}")
    assert e_corr < e_plain / 3


def test_persistent_grids_sized_for_fewer_cus_give_the_same_bits():
    """mb_set_cu_count (persistent grids on a CU-masked stream): 128 workgroups walk the tile list instead of 256 -- same tiles, same bits."""
    from maskbit_amd import _lib
    lib = _lib.load()
    torch.manual_seed(3)
    P, N, K = 4 * 257, 1024, 1024
    M = 2 * P
    A = torch.randn(M, K, device=DEV).half(); A[P:] *= 0.02
    W = (torch.randn(N, K, device=DEV) * 0.03).half()
    bias = torch.randn(N, device=DEV) * 0.1
    st = torch.cuda.current_stream().cuda_stream
    outs = []
    try:
        for n in (0, 128, 8):
            assert lib.mb_set_cu_count(n) == 0
            o = torch.empty(M, N, device=DEV, dtype=torch.float16)
            _lib.check(lib.mb_gemm_pair(0, A.data_ptr(), W.data_ptr(), bias.data_ptr(), None, None, o.data_ptr(), P, N, K, None, None, None, None, st))
            torch.cuda.synchronize()
            outs.append(o)
    finally:
        lib.mb_set_cu_count(0)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
