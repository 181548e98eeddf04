import sys, os, time, torch
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
from gemm_bench import run, dev
M = 128 * 257
for name, epi, N, K in [("ffn_down", 2, 1024, 4096)]:
    A = torch.randn(M, K, device=dev).half(); W = (torch.randn(N, K, device=dev) * 0.05).half()
    bias = torch.randn(N, device=dev) * 0.1; res = torch.randn(M, N, device=dev) if epi == 2 else None
    for rep in range(2):
        for v in (6, 16, 26, 8, 18, 28, 257):
            for _ in range(2): run(epi, A, W, bias, res, M, N, K, v)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): run(epi, A, W, bias, res, M, N, K, v)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
            print(f"{name} variant {v}: {dt*1e6:.1f} us", flush=True)
