"""Where does a trunk GEMM tile spend its time?  Sweeps K at fixed (M, N) for the plain sequence-aligned kernel and the CFG pair kernel and fits
time = tiles_per_CU * (fixed + K/64 * per_ktile): `per_ktile` is the steady-state K-loop cost of one 256x256x64 step, `fixed` what a tile pays
outside it (prologue wait, epilogue math, stores).  usage: python tools/k_sweep.py"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from maskbit_amd import _lib

dev = torch.device("cuda")
lib = _lib.load()
st = lambda: torch.cuda.current_stream().cuda_stream
ptr = lambda t: t.data_ptr() if t is not None else None


def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / n


def main():
    torch.manual_seed(0)
    P = 64 * 257
    M = 2 * P
    for name, epi, N in [("bias->h16 (qkv)", 0, 3072), ("gelu->h16 (ffn_up)", 1, 4096), ("residual->f32 (N=1024)", 2, 1024)]:
        for tag in ("plain", "pair"):
            ks, ts = [], []
            for K in (256, 512, 1024, 2048, 4096):
                A = torch.randn(M, K, device=dev).half()
                A[P:] *= 0.01
                W = (torch.randn(N, K, device=dev) * 0.05).half()
                bias = torch.randn(N, device=dev) * 0.1
                res = torch.randn(M, N, device=dev) if epi == 2 else None
                o32 = torch.empty(M, N, device=dev) if epi == 2 else None
                o16 = torch.empty(M, N, device=dev, dtype=torch.float16) if epi != 2 else None
                if tag == "plain":
                    fn = lambda: _lib.check(lib.mb_gemm(epi, A.data_ptr(), W.data_ptr(), bias.data_ptr(), ptr(res), ptr(o32), ptr(o16), M, N, K, 0, 257, st()))
                else:
                    fn = lambda: _lib.check(lib.mb_gemm_pair(epi, A.data_ptr(), W.data_ptr(), bias.data_ptr(), ptr(res), ptr(o32), ptr(o16), P, N, K, None, None, None, None, st()))
                dt = timeit(fn)
                ks.append(K // 64); ts.append(dt * 1e6)
                print(f"{name:24s} {tag:5s} K={K:5d}: {dt * 1e6:8.1f} us  {2.0 * M * N * K / dt / 1e12:7.1f} TFLOP/s", flush=True)
            rounds = (M // 257) * (N // 256) / 256.0                 # tiles per CU (128 m-tiles x N/256 over 256 CUs)
            slope, icpt = np.polyfit(np.array(ks, float), np.array(ts) / rounds, 1)
            print(f"  -> {rounds:.2f} tiles per CU; per tile: fixed {icpt:6.2f} us + {slope:5.3f} us per K-tile "
                  f"(MFMA-bound K-tile at 2.5 PFLOP/s: {2 * 272 * 256 * 64 / (2.5e15 / 256) * 1e6:.3f} us)", flush=True)


if __name__ == "__main__":
    main()
