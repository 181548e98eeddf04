"""Timing experiments on the trunk GEMM kernels (per-tile overhead vs per-K-tile cost, ablation variants)."""
import sys, os, time, torch
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
from gemm_bench import run, dev
M = 128 * 257
def timeit(epi, N, K, v, iters=10):
    A = torch.randn(M, K, device=dev).half(); W = (torch.randn(N, K, device=dev) * 0.05).half()
    bias = torch.randn(N, device=dev) * 0.1; res = torch.randn(M, N, device=dev) if epi == 2 else None
    for _ in range(3): run(epi, A, W, bias, res, M, N, K, v)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): run(epi, A, W, bias, res, M, N, K, v)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / iters * 1e6
if __name__ == "__main__":
  for epi, name in ((0, "h16"), (1, "gelu_h16"), (2, "res_f32")):
      for N in (1024, 3072):
          rounds = 128 * (N // 256) / 256
          ts = {K: timeit(epi, N, K, 257) for K in (512, 1024, 2048, 4096)}
          per_kt = (ts[4096] - ts[1024]) / rounds / 48
          ovh = ts[1024] / rounds - 16 * per_kt
          print(f"epi={name:8s} N={N}: " + " ".join(f"K={K}:{t:7.1f}us" for K, t in ts.items()) +
                f" | rounds={rounds:.0f} per-K-tile {per_kt:.3f} us, per-tile overhead {ovh:.2f} us", flush=True)
