"""Correctness + timing of the trunk GEMM family through the C ABI (mb_gemm), every kernel variant.
usage: python tools/gemm_bench.py [quick]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from maskbit_amd import _lib

dev = torch.device("cuda")
lib = _lib.load()


def run(epi, A, W, bias, res, M, N, K, variant, period=0):
    out32 = torch.empty(M if epi != 4 else (M // period) * (period - 1), N, device=dev, dtype=torch.float32) if epi in (2, 3, 4) else None
    out16 = torch.empty(M, N, device=dev, dtype=torch.float16) if epi in (0, 1) else None
    _lib.check(lib.mb_gemm(epi, A.data_ptr(), W.data_ptr(), bias.data_ptr(), res.data_ptr() if res is not None else None,
                           out32.data_ptr() if out32 is not None else None, out16.data_ptr() if out16 is not None else None,
                           M, N, K, period, variant, torch.cuda.current_stream().cuda_stream))
    return out32 if out32 is not None else out16


def reference(epi, A, W, bias, res, period=0):
    y = A.float() @ W.float().t() + bias
    if epi == 2: y = y + res
    if epi in (1, 3): y = torch.nn.functional.gelu(y)
    if epi == 4:
        M = y.shape[0]
        y = y.reshape(M // period, period, -1)[:, :period - 1].reshape(-1, y.shape[-1])
    return y


def main():  # noqa
    quick = "quick" in sys.argv
    torch.manual_seed(0)
    M = 128 * 257
    shapes = [("qkv", 0, 3072, 1024), ("attn_out", 2, 1024, 1024), ("ffn_up", 1, 4096, 1024), ("ffn_down", 2, 1024, 4096)]
    for name, epi, N, K in shapes:
        A = (torch.randn(M, K, device=dev) * 1.0).half()
        W = (torch.randn(N, K, device=dev) * 0.05).half()
        bias = torch.randn(N, device=dev) * 0.1
        res = torch.randn(M, N, device=dev) if epi == 2 else None
        ref = reference(epi, A[:4096], W, bias, res[:4096] if res is not None else None)
        ref_tail = reference(epi, A[-600:], W, bias, res[-600:] if res is not None else None)
        flops = 2.0 * M * N * K
        for variant in ([-1, 0] if quick else [-1, 6, 8, 257, 0]):
            out = run(epi, A, W, bias, res, M, N, K, variant)
            torch.cuda.synchronize()
            err = float((out[:4096].float() - ref).abs().max()); err2 = float((out[-600:].float() - ref_tail).abs().max())
            scale = float(ref.abs().max())
            for _ in range(2): run(epi, A, W, bias, res, M, N, K, variant)
            torch.cuda.synchronize()
            t0 = time.perf_counter(); n = 10
            for _ in range(n): run(epi, A, W, bias, res, M, N, K, variant)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
            print(f"{name:9s} N={N:5d} K={K:5d} variant={variant:2d}: {dt*1e6:8.1f} us  {flops/dt/1e12:7.1f} TFLOP/s  max_err={err:.4f}/{err2:.4f} (|ref|max {scale:.1f})", flush=True)


if __name__ == "__main__":
    main()
