"""Timing of the three ways a trunk GEMM can treat its activation operand: fp16 only, fp16 hi+lo (2 K sweeps), fp16 hi + e4m3 lo (1.5 sweeps)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maskbit_amd import _lib
lib = _lib.load(); dev = "cuda"
M = 128 * 257
st = lambda: torch.cuda.current_stream().cuda_stream
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for name, epi, N, K in (("qkv", 0, 3072, 1024), ("attn_out", 2, 1024, 1024), ("ffn_up", 1, 4096, 1024), ("ffn_down", 2, 1024, 4096)):
    A = torch.randn(M, K, device=dev).half(); Alo = (torch.randn(M, K, device=dev) * 1e-4).half()
    A8 = torch.zeros(M, 2 * K, device=dev, dtype=torch.uint8); A8[:, :K] = (Alo.float() * 2.0 ** 12).to(torch.float8_e4m3fn).view(torch.uint8)
    W = (torch.randn(N, K, device=dev) * 0.05).half()
    W8 = torch.zeros(N, 2 * K, device=dev, dtype=torch.uint8); W8[:, :K] = (W.float() * 2.0 ** 10).to(torch.float8_e4m3fn).view(torch.uint8)
    we = torch.tensor([10], device=dev, dtype=torch.int32)
    bias = torch.randn(N, device=dev) * 0.1
    res = torch.randn(M, N, device=dev) if epi == 2 else None
    o32 = torch.empty(M, N, device=dev) if epi == 2 else None; o16 = torch.empty(M, N, device=dev, dtype=torch.float16) if epi != 2 else None
    p = lambda t: t.data_ptr() if t is not None else None
    t0 = timeit(lambda: _lib.check(lib.mb_gemm(epi, p(A), p(W), p(bias), p(res), p(o32), p(o16), M, N, K, 0, 0, st())))
    t1 = timeit(lambda: _lib.check(lib.mb_gemm_act_split(epi, p(A), p(Alo), p(W), p(bias), p(res), p(o32), p(o16), M, N, K, 0, st())))
    t2 = timeit(lambda: _lib.check(lib.mb_gemm_f8lo(epi, p(A), p(A8), p(W), p(W8), p(we), p(bias), p(res), p(o32), p(o16), M, N, K, 0, st())))
    print(f"{name:9s}: fp16 {t0:7.1f} us   hi+lo fp16 {t1:7.1f} us ({t1 / t0:.2f}x)   hi + e4m3 lo {t2:7.1f} us ({t2 / t0:.2f}x)", flush=True)
