import sys, os, time, torch
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
from gemm_bench import run, dev
M = 128 * 257
def timeit(epi, N, K, v, iters=20):
    A = torch.randn(M, K, device=dev).half(); W = (torch.randn(N, K, device=dev) * 0.05).half()
    bias = torch.randn(N, device=dev) * 0.1; res = torch.randn(M, N, device=dev) if epi == 2 else None
    for _ in range(3): run(epi, A, W, bias, res, M, N, K, v)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): run(epi, A, W, bias, res, M, N, K, v)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / iters * 1e6
for rep in range(3):
    for name, epi, N, K in (("attn_out", 2, 1024, 1024), ("ffn_down", 2, 1024, 4096)):
        print(name, " ".join(f"v{v}: {timeit(epi, N, K, v):.1f}us" for v in (257, 6, 8, -1)), flush=True)
