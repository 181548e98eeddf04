"""Timing of the CFG pair GEMM (mb_gemm_pair) next to the plain sequence-aligned GEMM on the four trunk shapes (B = 64 pairs), with and without
the MX-fp4 weight-correction pass; plus the apples-to-apples 8192^3 number of the plain kernel.
usage: python tools/pair_bench.py"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from maskbit_amd import _lib

dev = torch.device("cuda")
lib = _lib.load()
st = lambda: torch.cuda.current_stream().cuda_stream
ptr = lambda t: t.data_ptr() if t is not None else None


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def main():
    torch.manual_seed(0)
    P = 64 * 257
    M = 2 * P
    for name, epi, N, K in [("qkv", 0, 3072, 1024), ("attn_out", 2, 1024, 1024), ("ffn_up", 1, 4096, 1024), ("ffn_down", 2, 1024, 4096)]:
        A = torch.randn(M, K, device=dev).half()
        A[P:] *= 0.01                                     # difference rows are small
        W = (torch.randn(N, K, device=dev) * 0.05).half()
        W32 = W.float()
        bias = torch.randn(N, device=dev) * 0.1
        res = torch.randn(M, N, device=dev) if epi == 2 else None
        o32 = torch.empty(M, N, device=dev) if epi == 2 else None
        o16 = torch.empty(M, N, device=dev, dtype=torch.float16) if epi != 2 else None
        x4 = torch.randint(0, 256, (M, 2 * K), device=dev, dtype=torch.uint8)
        xs = torch.full((M + 256,), 100, device=dev, dtype=torch.uint8); xs[P:] = 0                    # plain tiles: one byte per row
        xsb = torch.full((P * (K // 64) + 256,), 100, device=dev, dtype=torch.uint8)                    # pair tiles: one byte per (row, 64 K-elements)
        w4 = torch.zeros(N, 2 * K, device=dev, dtype=torch.uint8); ws = torch.zeros(N, device=dev, dtype=torch.uint8)
        _lib.check(lib.mb_w4_from_f32(W32.data_ptr(), N, K, w4.data_ptr(), ws.data_ptr(), st()))
        flops = 2.0 * M * N * K
        plain = lambda: _lib.check(lib.mb_gemm(epi, A.data_ptr(), W.data_ptr(), bias.data_ptr(), ptr(res), ptr(o32), ptr(o16), M, N, K, 0, 257, st()))
        pair = lambda: _lib.check(lib.mb_gemm_pair(epi, A.data_ptr(), W.data_ptr(), bias.data_ptr(), ptr(res), ptr(o32), ptr(o16), P, N, K, None, None, None, None, st()))
        pair4 = lambda: _lib.check(lib.mb_gemm_pair(epi, A.data_ptr(), W.data_ptr(), bias.data_ptr(), ptr(res), ptr(o32), ptr(o16), P, N, K,
                                                    x4.data_ptr(), xsb.data_ptr(), w4.data_ptr(), ws.data_ptr(), st()))
        f4 = lambda: _lib.check(lib.mb_gemm_f4lo(epi, A.data_ptr(), x4.data_ptr(), xs.data_ptr(), W.data_ptr(), w4.data_ptr(), ws.data_ptr(), bias.data_ptr(),
                                                 ptr(res), ptr(o32), ptr(o16), M, N, K, 257, st()))
        for tag, fn in (("plain", plain), ("pair", pair), ("pair+f4", pair4), ("plain+f4", f4)):
            dt = timeit(fn)
            print(f"{name:9s} {tag:9s}: {dt * 1e6:8.1f} us  {flops / dt / 1e12:7.1f} TFLOP/s (algorithmic)", flush=True)
    if "big" in sys.argv:
        n = 8192
        A = torch.randn(n, n, device=dev).half(); W = torch.randn(n, n, device=dev).half(); bias = torch.zeros(n, device=dev)
        o16 = torch.empty(n, n, device=dev, dtype=torch.float16)
        for v in (8, 0):
            dt = timeit(lambda: _lib.check(lib.mb_gemm(0, A.data_ptr(), W.data_ptr(), bias.data_ptr(), None, None, o16.data_ptr(), n, n, n, 0, v, st())), n=10)
            print(f"8192^3 fp16, N(0,1) operands, variant {v}: {dt * 1e6:8.1f} us  {2.0 * n ** 3 / dt / 1e12:7.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
