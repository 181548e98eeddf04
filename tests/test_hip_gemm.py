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
    _lib.check(lib.mb_gemm_ex(2, A.data_ptr(), W.data_ptr(), bias.data_ptr(), buf.data_ptr(), buf.data_ptr(), None, M, N, K,
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
    """fp16 hi+lo activation pairs (the plain forward's LayerNorm outputs at mb_gen_cfg.precision >= 1): mb_layernorm writes x_hi and x_lo = fp16(x - x_hi); the GEMM over the pair
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


# (the MX-fp4 mini-tile passes: tests/test_hip_mini.py)


def test_persistent_grids_sized_for_fewer_cus_give_the_same_bits():
    from hip_helpers import gemm_mini
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
            gemm_mini(lib, 0, A, W, bias, None, None, o, P, True, N, K)
            torch.cuda.synchronize()
            outs.append(o)
    finally:
        lib.mb_set_cu_count(0)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
