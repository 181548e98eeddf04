"""Is the fp32+residual epilogue bound per CU or by aggregate HBM bandwidth?  Time the out-proj GEMM (N = K = 1024) at
M = 257 * nb for nb sequences: tiles = 4 * nb, one tile per workgroup, one workgroup per CU."""
import sys, time, torch
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
from gemm_bench import run, dev
def timeit(epi, M, N, K, iters=30):
    A = torch.randn(M, K, device=dev).half(); W = (torch.randn(N, K, device=dev) * 0.05).half()
    bias = torch.randn(N, device=dev) * 0.1; res = torch.randn(M, N, device=dev) if epi == 2 else None
    for _ in range(3): run(epi, A, W, bias, res, M, N, K, 257)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): run(epi, A, W, bias, res, M, N, K, 257)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / iters * 1e6
for K in (1024, 4096):
    for nb in (8, 16, 32, 64, 128):
        t2 = timeit(2, 257 * nb, 1024, K); t0 = timeit(0, 257 * nb, 1024, K)
        print(f"K={K} nb={nb:3d} tiles={4 * nb:4d}: fp32+residual {t2:7.1f} us   fp16 {t0:7.1f} us   diff {t2 - t0:6.1f} us", flush=True)
